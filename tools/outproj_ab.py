#!/usr/bin/env python3
"""The decoder's out-projection input gradient [rows x 512 x 512] (bf16 store): the streaming 4 x 128-column kernel (policy 1) against the 256 x 256 tile on the 8-phase
K loop (policy 2), interleaved rounds in one process.  python tools/outproj_ab.py   (one MI355X)"""
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


for M, lim in ((61519, None), (81920, 61519), (81920, None)):
	a = (torch.rand(M, 512, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(512, 512, device="cuda") * 2 - 1).to(torch.bfloat16)
	out = torch.empty(M, 512, dtype=torch.bfloat16, device="cuda")
	rl = None if lim is None else torch.tensor([lim], dtype=torch.int32, device="cuda")
	fn = lambda: ops.gemm(a, b, M, 512, 512, out=out, row_limit=rl)
	res = {1: [], 2: []}
	for pol in (1, 2):
		ops.gemm_tile_policy(pol)
		for _ in range(3):
			fn()
		tile = ops.gemm_last_tile()
		res[pol].append(tile)
	torch.cuda.synchronize()
	tiles = {p: res[p].pop() for p in res}
	for rnd in range(7):
		for pol in (1, 2):
			ops.gemm_tile_policy(pol)
			res[pol].append(time_once(fn))
	ops.gemm_tile_policy(1)
	print(f"M={M} row_limit={lim}: policy 1 (tile {tiles[1]}) {statistics.median(res[1]):6.1f} us | policy 2 (tile {tiles[2]}) {statistics.median(res[2]):6.1f} us")
