#!/usr/bin/env python3
"""K-contiguous x K-contiguous GEMM sweep under the tile policies 0 (128^2), 2 (256-wide LDS-DMA tile), 3 (192-wide) -- run on the GPU box:
   python tools/gemm_sweep.py [--resid] [M,N,K ...]      --resid: RESID_F32 epilogue (fp32 residual in, fp32 out) instead of STORE_BF16"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402

resid = "--resid" in sys.argv
drop = ops.Dropout(0.1, 123, 5) if "--drop" in sys.argv else ops.NO_DROPOUT  # with --resid: dropout on the GEMM output, as in training
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:] if not a.startswith("--")] or [(4096, 4096, 4096), (8192, 8192, 8192), (57344, 6912, 512), (57344, 6912, 2048), (81920, 1536, 512)]
for m, n, k in shapes:
	a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
	out = torch.empty(m, n, dtype=torch.float32 if resid else torch.bfloat16, device="cuda")
	rs = torch.randn(m, n, device="cuda") if resid else None
	kw = dict(kind=ops.EPI_RESID_F32, resid=rs, dropout=drop) if resid else {}
	res, ref = [], None
	for pol in (0, 2, 3):
		ops.gemm_tile_policy(pol)
		for _ in range(3):
			ops.gemm(a, b, m, n, k, out=out, **kw)
		if ref is None:
			ref = out.clone()
		else:
			assert torch.equal(ref, out), f"policy {pol} differs from the 128^2 kernel"
		s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		torch.cuda.synchronize()
		s.record()
		reps = 10
		for _ in range(reps):
			ops.gemm(a, b, m, n, k, out=out, **kw)
		e.record()
		torch.cuda.synchronize()
		us = s.elapsed_time(e) / reps * 1000
		res.append((us, 2.0 * m * n * k / us / 1e6))
	print(f"M={m:6d} N={n:5d} K={k:5d}   128^2: {res[0][0]:8.1f} us {res[0][1]:7.1f} TF   256x256: {res[1][0]:8.1f} us {res[1][1]:7.1f} TF   256x192: {res[2][0]:8.1f} us {res[2][1]:7.1f} TF",
	      flush=True)
ops.gemm_tile_policy(1)
