#!/usr/bin/env python3
"""192 x 256 tiles (gemm256p_kernel<RESID_F32, 6>) against 256 x 256 tiles: the towers' fp32-residual GEMMs whose 256-row tiles fill less than a round (isolated, interleaved),
then the ViT-B/32 and text towers whole, one tower captured under each setting (novic_gemm256_pipeline(10 / 11)).  python tools/tile192_ab.py"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import clip_text, clip_vit, ops  # noqa: E402


def time_once(fn, n=20):
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for _ in range(n):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / n * 1000


prev = ops.gemm_tile_policy(2)  # (isolated part: the 256-wide kernels for every shape, the text tower's [rows x 512 x 512] included)
for name, M, N, K in (("ViT-B/32 proj", 12800, 768, 768), ("ViT-B/32 fc2", 12800, 768, 3072), ("text proj", 19712, 512, 512), ("text fc2", 19712, 512, 2048)):
	a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	b = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	out = torch.empty(M, N, device="cuda")
	kw = dict(kind=ops.EPI_RESID_F32, resid=torch.randn(M, N, device="cuda"), bias=torch.randn(N, device="cuda"), split_tail=True)
	fn = lambda: ops.gemm(a, b, M, N, K, out=out, **kw)
	res, plans = {10: [], 11: []}, {}
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
	fl = 2.0 * M * N * K
	print(f"{name:14s} [{M} x {N} x {K}]: " + " | ".join(f"{'192-row' if p == 11 else '256-row'} tiles ({plans[p]['workgroups']} workgroups) {statistics.median(v):6.1f} us {fl / statistics.median(v) / 1e6:5.0f} TF"
	                                                    for p, v in res.items()), flush=True)
ops.gemm_tile_policy(prev)

with torch.no_grad():
	for label, make, x, unit in (("ViT-B/32 batch 256", lambda: clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).cuda(), torch.randn(256, 3, 224, 224).cuda(), "img/s"),
	                             ("text tower batch 256", lambda: clip_text.NativeTextTower(clip_text.TEXT_B_32, seed=3).cuda(), None, "texts/s")):
		if x is None:
			x = torch.randint(1, 49000, (256, 77)).cuda()
			x[:, -1] = 49407
		towers, outs = {}, {}
		for pol in (10, 11):
			ops.gemm256_pipeline(pol)
			towers[pol] = make()
			for _ in range(4):
				outs[pol] = towers[pol](x)  # (eager, capture, replay: the graph holds the kernels chosen under this setting)
		torch.cuda.synchronize()
		res = {10: [], 11: []}
		for rnd in range(7):
			for pol, t in towers.items():
				torch.cuda.synchronize()
				t0 = time.perf_counter()
				for _ in range(5):
					t(x)
				torch.cuda.synchronize()
				res[pol].append((time.perf_counter() - t0) / 5)
		print(f"{label}: " + " | ".join(f"{'192-row' if p == 11 else '256-row'} tiles {statistics.median(v) * 1e3:.3f} ms, {256 / statistics.median(v):.0f} {unit}" for p, v in res.items())
		      + f" | bit-identical outputs: {bool(torch.equal(outs[10], outs[11]))}", flush=True)
ops.gemm256_pipeline(10)
