#!/usr/bin/env python3
"""A/B of a library switch INSIDE the training step (bench.py's step, interleaved rounds in one process): python tools/step_ab.py pipeline 0 1
(novic_gemm256_pipeline: one barrier per K-tile / the 8-phase K loop); python tools/step_ab.py wgrad 0 1; python tools/step_ab.py attr:prefix_wgrad256 0 1."""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import _lib, embedding_noise, train as T  # noqa: E402

which, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if which.startswith("attr:"):  # a class attribute of the decoder (e.g. attr:prefix_wgrad256 0 1)
	from novic_amd import embedding_decoder
	setter = lambda v: setattr(embedding_decoder.PrefixedIterDecoder, which[5:], bool(v))
else:
	setter = {"tile_policy": _lib.lib().novic_gemm_tile_policy, "pipeline": _lib.lib().novic_gemm256_pipeline, "wgrad": _lib.lib().novic_wgrad_policy}[which]
dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.1, device=dev)
model.train()
opt = T.FusedAdamW(model, lr=1.5e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
noise = embedding_noise.EmbeddingNoise.create("GaussElemUniformAngle", bench.F_DIM, 3.25, 45.0, 75.0, 0.0, 0.15)
mbs = [bench.synth_micro_batch(spec, bench.MICRO_B, 100 + j, dev) for j in range(bench.ACCUM)]
step = lambda: T.train_step(model, opt, [(e.clone(), t, p, w) for e, t, p, w in mbs], embed_noise=noise)
for _ in range(4):
	step()
res = {a: [], b: []}
for rnd in range(9):
	for v in (a, b):
		setter(v)
		step()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(6):
			step()
		torch.cuda.synchronize()
		res[v].append((time.perf_counter() - t0) / 6)
for v in (a, b):
	print(f"{which}({v}): {statistics.median(res[v]) * 1e3:.3f} ms per step (min {min(res[v]) * 1e3:.3f})", flush=True)
