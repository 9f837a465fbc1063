#!/usr/bin/env python3
"""Race screen of the 8-phase weight-gradient kernel: the same launch N times against the one-barrier kernel's bits, per shape; prints how many repetitions differed and how
(number of differing elements, their tile, max |d|).  python tools/wgrad_race_screen.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
for M, N, K in ((6912, 512, 36943), (1536, 512, 61519), (512, 512, 61519)):
	g = torch.Generator().manual_seed(M * 7 + N + K)
	dy = (torch.randn(K, M, generator=g) * 0.5).to(torch.bfloat16).cuda()
	x = (torch.randn(K, N, generator=g) * 0.5).to(torch.bfloat16).cuda()
	ops.wgrad_policy(0)
	ref = torch.zeros(M, N, device="cuda")
	ops.wgrad(dy, x, M, N, K, ref)
	ops.wgrad_policy(1)
	bad = []
	keep = []
	for r in range(reps):
		out = torch.zeros(M, N, device="cuda")  # (a fresh output per repetition, twelve kept alive, no synchronisation in between: as the test that once failed does it)
		keep.append(out)
		if len(keep) > 12:
			keep.pop(0)
		ops.wgrad(dy, x, M, N, K, out)
		if r % 12 != 11:
			continue
		torch.cuda.synchronize()
		for out in keep:
			d = (out - ref)
			nz = d.nonzero()
			if nz.numel():
				rows, cols = nz[:, 0], nz[:, 1]
				bad.append((r, int(nz.shape[0]), int(rows.min()), int(rows.max()), int(cols.min()), int(cols.max()), float(d.abs().max())))
		continue
		d = (out - ref)
		nz = d.nonzero()
		if nz.numel():
			rows, cols = nz[:, 0], nz[:, 1]
			bad.append((r, int(nz.shape[0]), int(rows.min()), int(rows.max()), int(cols.min()), int(cols.max()), float(d.abs().max())))
	print(f"[{M} x {N}] K {K}: {len(bad)} of {reps} repetitions differ", bad[:6], flush=True)
