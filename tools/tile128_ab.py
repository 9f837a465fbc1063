#!/usr/bin/env python3
"""128 x 256 tiles (gemm256p_kernel<EPI, 4>) against 256 x 256 tiles on the towers' fp32-residual GEMMs whose 256-row tiles fill less than a round of the chip, interleaved
rounds in one process: python tools/tile128_ab.py [rows] [width]   (defaults: ViT-B/32 at batch 256 = 12 800 rows, width 768)"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
W = int(sys.argv[2]) if len(sys.argv) > 2 else 768


def time_once(fn, n=20):
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for _ in range(n):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / n * 1000


for name, N, K in (("proj", W, W), ("fc2", W, 4 * W)):
	a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	out = torch.empty(M, N, device="cuda")
	kw = dict(kind=ops.EPI_RESID_F32, resid=torch.randn(M, N, device="cuda"), bias=torch.randn(N, device="cuda"), split_tail=True)
	fn = lambda: ops.gemm(a, b, M, N, K, out=out, **kw)
	res = {6: [], 7: []}
	plans = {}
	for pol in res:
		ops.gemm256_pipeline(pol)
		plans[pol] = ops.gemm256_plan(M, N, K, kind=ops.EPI_RESID_F32, bias=True, split_tail=True)
		for _ in range(3):
			fn()
	torch.cuda.synchronize()
	for rnd in range(7):
		for pol in res:
			ops.gemm256_pipeline(pol)
			res[pol].append(time_once(fn))
	ops.gemm256_pipeline(7)
	fl = 2.0 * M * N * K
	print(f"{name:5s} [{M} x {N} x {K}]: " + " | ".join(f"{'128-row' if p == 7 else '256-row'} tiles {plans[p]} {statistics.median(v):6.1f} us {fl / statistics.median(v) / 1e6:5.0f} TF" for p, v in res.items()), flush=True)
