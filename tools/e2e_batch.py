#!/usr/bin/env python3
"""End-to-end ViT-B/32 + greedy / beam-4 decode as a function of the batch (the reference uses ONE batch size for the image tower and the decoder: infer.py:99-101): tower alone,
decode alone, one after the other, pipelined (embedders.pipeline_image_batches).  python tools/e2e_batch.py [batches ...] (default 256 512 1024)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import clip_vit, embedders  # noqa: E402

dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
vit = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).to(dev)


def rate(fn, n, reps=6):
	with torch.no_grad():
		for _ in range(3):
			fn()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(reps):
			fn()
		torch.cuda.synchronize()
	return n * reps / (time.perf_counter() - t0)


for B in [int(a) for a in sys.argv[1:]] or [256, 512, 1024]:
	g = torch.Generator().manual_seed(B)
	seq = [torch.randn(B, 3, 224, 224, generator=g).to(dev) for _ in range(4)]
	emb = torch.nn.functional.normalize(torch.randn(B, spec.embed_dim, generator=g), dim=-1).to(dev)
	greedy = lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)
	beam4 = lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)
	line = [f"batch {B}: tower {rate(lambda: vit(seq[0]), B) / 1e3:.1f} k images/s"]
	for name, dec in (("greedy", greedy), ("beam-4", beam4)):
		def piped():
			for e in embedders.pipeline_image_batches(vit, seq, dev):
				dec(e)
		line.append(f"{name}: decode alone {rate(lambda: dec(emb), B) / 1e3:.1f} k, one after the other {rate(lambda: dec(vit(seq[0])), B) / 1e3:.1f} k, "
		            f"pipelined {rate(piped, B * len(seq), reps=3) / 1e3:.1f} k labels/s")
	print(" | ".join(line), flush=True)
	del seq
