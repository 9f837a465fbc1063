#!/usr/bin/env python3
"""A/B of the output-store policy of the 256-wide GEMM tiles' bf16 outputs (novic_epilogue_t.store_policy; here through the process default, novic_gemm256_pipeline(4 / 5)):
non-temporal (streamed past L2: measured best for the training step's huge GEMMs, whose operand panels must survive in L2) against ordinary write-back stores (a tower's
activations -- qkv 59 MB, hid 79 MB at ViT-B/32 batch 256 -- can then be served to the next kernel out of L2 / the 256 MB Infinity Cache).  Towers: each variant captures
its own graphs.  Training step: bench.py's step.  python tools/store_policy_ab.py"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import _lib, clip_text, clip_vit, embedding_noise, siglip, train as T  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda")


def ab(name, make, x, per, unit):
	towers = {}
	with torch.no_grad():
		for pol in (4, 5):
			L.novic_gemm256_pipeline(pol)
			t = make()
			for _ in range(4):
				o = t(x)
			towers[pol] = (t, o.clone())
		torch.cuda.synchronize()
		res = {4: [], 5: []}
		for rnd in range(7):
			for pol, (t, _) in towers.items():
				torch.cuda.synchronize()
				t0 = time.perf_counter()
				for _ in range(5):
					t(x)
				torch.cuda.synchronize()
				res[pol].append((time.perf_counter() - t0) / 5)
	L.novic_gemm256_pipeline(4)
	same = bool(torch.equal(towers[4][1], towers[5][1]))
	print(f"{name}: non-temporal {statistics.median(res[4]) * 1e3:.3f} ms ({per / statistics.median(res[4]):.0f} {unit}) | write-back {statistics.median(res[5]) * 1e3:.3f} ms "
	      f"({per / statistics.median(res[5]):.0f} {unit}); bit-identical outputs: {same}", flush=True)


B = 256
ab("ViT-B/32 batch 256", lambda: clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).cuda(), torch.randn(B, 3, 224, 224).cuda(), B, "img/s")
ab("text B/32 batch 256", lambda: clip_text.NativeTextTower(clip_text.TEXT_B_32, seed=3).cuda(), torch.randint(1, 49406, (B, 77)).cuda(), B, "texts/s")
ab("SigLIP B/16 batch 256", lambda: siglip.NativeSigLIPViT(siglip.SigLIPVisionConfig(224, 16, 768, 12, 12, 3072), seed=3).cuda(), torch.randn(B, 3, 224, 224).cuda(), B, "img/s")
ab("ViT-L/14 batch 256", lambda: clip_vit.NativeViT(clip_vit.VIT_L_14, seed=3).cuda(), torch.randn(B, 3, 224, 224).cuda(), B, "img/s")

# training step
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.1, device=dev)
model.train()
opt = T.FusedAdamW(model, lr=1.5e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
noise = embedding_noise.EmbeddingNoise.create("GaussElemUniformAngle", bench.F_DIM, 3.25, 45.0, 75.0, 0.0, 0.15)
mbs = [bench.synth_micro_batch(spec, bench.MICRO_B, 100 + j, dev) for j in range(bench.ACCUM)]
step = lambda: T.train_step(model, opt, [(e.clone(), t, p, w) for e, t, p, w in mbs], embed_noise=noise)
for _ in range(3):
	step()
res = {4: [], 5: []}
for rnd in range(7):
	for pol in res:
		L.novic_gemm256_pipeline(pol)
		step()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(4):
			step()
		torch.cuda.synchronize()
		res[pol].append((time.perf_counter() - t0) / 4)
L.novic_gemm256_pipeline(4)
print(f"training step: non-temporal {statistics.median(res[4]) * 1e3:.3f} ms | write-back {statistics.median(res[5]) * 1e3:.3f} ms", flush=True)
