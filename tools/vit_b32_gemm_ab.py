#!/usr/bin/env python3
"""ViT-B/32 at batch 256 (12 800 rows): the fp32-residual GEMMs (proj, fc2) on the 256 x 192 tile (policy 1's choice: 200 tiles, one-barrier K loop) against the 256 x 256
tile on the 8-phase K loop (policy 2: 150 tiles), and QKV / fc1 for reference; interleaved rounds.  python tools/vit_b32_gemm_ab.py [rows] [width]"""
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


for name, N, K, mode in (("proj", W, W, "resid"), ("fc2", W, 4 * W, "resid"), ("qkv", 3 * W, W, "bias"), ("fc1", 4 * W, W, "qgelu")):
	a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	bias = torch.randn(N, device="cuda")
	if mode == "resid":
		out = torch.empty(M, N, device="cuda")
		kw = dict(kind=ops.EPI_RESID_F32, resid=torch.randn(M, N, device="cuda"), bias=bias, split_tail=True)
	else:
		out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
		kw = dict(bias=bias, act=ops.ACT_QUICKGELU if mode == "qgelu" else ops.ACT_NONE, split_tail=True)
	fn = lambda: ops.gemm(a, b, M, N, K, out=out, **kw)
	res, tiles = {0: [], 1: [], 2: [], 3: []}, {}
	for pol in res:
		ops.gemm_tile_policy(pol)
		for _ in range(3):
			fn()
		tiles[pol] = ops.gemm_last_tile()
	torch.cuda.synchronize()
	for rnd in range(7):
		for pol in res:
			ops.gemm_tile_policy(pol)
			res[pol].append(time_once(fn))
	ops.gemm_tile_policy(1)
	fl = 2.0 * M * N * K
	print(f"{name:5s} [{M} x {N} x {K}]: " + " | ".join(f"policy {p} (tile {tiles[p]}) {statistics.median(v):6.1f} us {fl / statistics.median(v) / 1e6:5.0f} TF" for p, v in res.items()), flush=True)
