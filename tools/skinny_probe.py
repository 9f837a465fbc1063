#!/usr/bin/env python3
"""The feed-forward block's narrow GEMMs under tile policy 0 (128^2 kernel) and 1 (skinny.hip) -- run on the GPU box: python tools/skinny_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402


def timeit(fn, reps=20):
	for _ in range(3):
		fn()
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	torch.cuda.synchronize()
	s.record()
	for _ in range(reps):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) * 1000 / reps


M = 81920
d = ops.Dropout(0.1, 5, 3)
a = (torch.rand(M, 512, device="cuda") - 0.5).bfloat16()
w = (torch.rand(128, 512, device="cuda") - 0.5).bfloat16()
y, y2 = torch.empty(M, 128, device="cuda", dtype=torch.bfloat16), torch.empty(M, 128, device="cuda", dtype=torch.bfloat16)
a2 = (torch.rand(M, 128, device="cuda") - 0.5).bfloat16()
w2 = (torch.rand(512, 128, device="cuda") - 0.5).bfloat16()
r, o = torch.randn(M, 512, device="cuda"), torch.empty(M, 512, device="cuda")
for pol in (0, 1):
	ops.gemm_tile_policy(pol)
	print("policy", pol,
	      "| [M x 128 x 512] GELU+drop %.1f us" % timeit(lambda: ops.gemm(a, w, M, 128, 512, kind=ops.EPI_GELU_BF16, out=y, out2=y2, dropout=d)),
	      "GELU'+drop %.1f us" % timeit(lambda: ops.gemm(a, w, M, 128, 512, kind=ops.EPI_GELU_BWD_BF16, out=y, resid=y2, dropout=d)),
	      "store %.1f us" % timeit(lambda: ops.gemm(a, w, M, 128, 512, out=y)),
	      "| [M x 512 x 128] resid+drop %.1f us" % timeit(lambda: ops.gemm(a2, w2, M, 512, 128, kind=ops.EPI_RESID_F32, out=o, resid=r, dropout=d)), flush=True)
ops.gemm_tile_policy(1)
