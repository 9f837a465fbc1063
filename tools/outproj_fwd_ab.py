#!/usr/bin/env python3
"""The decoder's out-projection + residual [rows x 512 x 512], fp32 residual epilogue with dropout, device row count (packed rows): the streaming 4 x 128-column kernel
(policy 1: skinny_n128_kernel<3, 2>, 70-78 us in the step) against the 256 x 256 tile on the 8-phase K loop (policy 2), interleaved rounds in one process; bit-identity
of the two outputs.  python tools/outproj_fwd_ab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402


def time_once(fn, n=20):
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for _ in range(n):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / n * 1000


for M, lim, p in ((81920, 61519, 0.1), (81920, 61519, 0.0), (61519, None, 0.1)):
	a = (torch.rand(M, 512, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(512, 512, device="cuda") * 2 - 1).to(torch.bfloat16)
	resid = torch.randn(M, 512, device="cuda")
	rl = None if lim is None else torch.tensor([lim], dtype=torch.int32, device="cuda")
	outs = {}
	res = {1: [], 2: []}
	tiles = {}
	for pol in (1, 2):
		ops.gemm_tile_policy(pol)
		out = torch.zeros(M, 512, device="cuda")
		fn = lambda o=out: ops.gemm(a, b, M, 512, 512, kind=ops.EPI_RESID_F32, out=o, resid=resid, row_limit=rl, dropout=ops.Dropout(p, seed=7, site=3))
		for _ in range(3):
			fn()
		tiles[pol] = ops.gemm_last_tile()
		outs[pol] = (out, fn)
	torch.cuda.synchronize()
	for rnd in range(7):
		for pol in (1, 2):
			ops.gemm_tile_policy(pol)
			res[pol].append(time_once(outs[pol][1]))
	ops.gemm_tile_policy(1)
	rows = lim or M
	same = bool(torch.equal(outs[1][0][:rows], outs[2][0][:rows]))
	print(f"M={M} row_limit={lim} dropout={p}: policy 1 (tile {tiles[1]}) {statistics.median(res[1]):6.1f} us | policy 2 (tile {tiles[2]}) {statistics.median(res[2]):6.1f} us; bit-identical: {same}", flush=True)
