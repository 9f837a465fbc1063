#!/usr/bin/env python3
"""Probe: would two half-steps on streams of their own, each with half the CUs for its persistent GEMM grids, beat one full step?  (The step is ~2.9 ms of MFMA-class
kernels + ~3.5 ms of HBM-class kernels run one after the other; two lanes could run one lane's HBM-bound phases under the other's K loops.)  Two MODEL INSTANCES, so no
gradient buffer is shared: this measures the hardware question only.  python tools/train_lanes_probe.py"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import embedding_noise, ops, train as T  # noqa: E402

dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
ACC = bench.ACCUM


def make(seed):
	torch.manual_seed(seed)
	m = bench.build_decoder(spec, dropout=0.1, device=dev)
	m.train()
	opt = T.FusedAdamW(m, lr=1.5e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	noise = embedding_noise.EmbeddingNoise.create("GaussElemUniformAngle", bench.F_DIM, 3.25, 45.0, 75.0, 0.0, 0.15)
	return m, opt, noise


full = make(0)
halves = [make(1), make(2)]
mbs = [bench.synth_micro_batch(spec, bench.MICRO_B, 100 + j, dev) for j in range(ACC)]


def step(pack, batches):
	m, opt, noise = pack
	fresh = [(e.clone(), t, p, w) for e, t, p, w in batches]
	return T.train_step(m, opt, fresh, embed_noise=noise)


streams = [torch.cuda.Stream(), torch.cuda.Stream()]
main = torch.cuda.current_stream()


def two_lanes(cus):
	for i in range(2):
		streams[i].wait_stream(main)
		with torch.cuda.stream(streams[i]), ops.cu_budget(cus):
			step(halves[i], mbs[i * ACC // 2:(i + 1) * ACC // 2])
	for s in streams:
		main.wait_stream(s)


variants = {"one step, 16 micro-batches, 256 CUs": lambda: step(full, mbs),
            "one half step (8 micro-batches) alone, 256 CUs": lambda: step(halves[0], mbs[:ACC // 2]),
            "two half steps one after the other": lambda: (step(halves[0], mbs[:ACC // 2]), step(halves[1], mbs[ACC // 2:]))}
for cus in (256, 160, 128, 96):
	variants[f"two half steps concurrently, GEMM grids on {cus} CUs each"] = (lambda c: (lambda: two_lanes(c)))(cus)
res = {k: [] for k in variants}
host = {k: [] for k in variants}
for k, fn in variants.items():
	for _ in range(3):
		fn()
torch.cuda.synchronize()
for rnd in range(5):
	for k, fn in variants.items():
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(4):
			fn()
		t1 = time.perf_counter()
		torch.cuda.synchronize()
		res[k].append((time.perf_counter() - t0) / 4)
		host[k].append((t1 - t0) / 4)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
for k in variants:
	print(f"{k:62s}: {statistics.median(res[k]) * 1e3:7.3f} ms   (host time to enqueue: {statistics.median(host[k]) * 1e3:6.3f} ms)", flush=True)
