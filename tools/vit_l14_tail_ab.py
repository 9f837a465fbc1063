#!/usr/bin/env python3
"""ViT-L/14 at batch 256 (65 792 rows): QKV [.. x 3072 x 1024] = 3084 tiles and fc1 [.. x 4096 x 1024] = 4112 tiles of 256 x 256 -- whole rounds of 256 plus 12 / 16 tiles.
The K-split of those tail tiles (4 parts each) against an extra round for them; interleaved rounds.  python tools/vit_l14_tail_ab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402


def time_once(fn, n=5):
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for _ in range(n):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / n * 1000


M = 65792
for name, N, K, act in (("qkv", 3072, 1024, ops.ACT_NONE), ("fc1", 4096, 1024, ops.ACT_GELU)):
	a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	bias = torch.randn(N, device="cuda")
	out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
	fn = lambda: ops.gemm(a, b, M, N, K, out=out, bias=bias, act=act, split_tail=True)
	res = {2: [], 3: []}
	for knob in res:
		ops.gemm256_pipeline(knob)
		for _ in range(2):
			fn()
	torch.cuda.synchronize()
	for rnd in range(6):
		for knob in res:
			ops.gemm256_pipeline(knob)
			res[knob].append(time_once(fn))
	ops.gemm256_pipeline(3)
	fl = 2.0 * M * N * K
	print(f"{name} [{M} x {N} x {K}]: extra round {statistics.median(res[2]):7.1f} us ({fl / statistics.median(res[2]) / 1e6:5.0f} TF) | K-split tail {statistics.median(res[3]):7.1f} us "
	      f"({fl / statistics.median(res[3]) / 1e6:5.0f} TF)", flush=True)
