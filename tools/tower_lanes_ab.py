#!/usr/bin/env python3
"""A tower with its batch on one stream against sub-batches on streams of their own (NativeViT.lanes; whole forward replayed from one hipGraph with a branch per lane),
the lanes' persistent GEMM grids on disjoint CU budgets (NativeViT.lane_cus), interleaved rounds: python tools/tower_lanes_ab.py [CFG] [batch]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import clip_vit  # noqa: E402

cfg = getattr(clip_vit, sys.argv[1] if len(sys.argv) > 1 else "VIT_B_32")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = torch.randn(B, 3, cfg.image_size, cfg.image_size).cuda()
variants = [(1, None), (2, 128), (2, 160), (2, 256), (3, 88), (4, 64)]
towers = {}
for v in variants:
	t = clip_vit.NativeViT(cfg, seed=3).cuda()
	t.lanes, t.lane_cus, t.lane_min_rows = v[0], v[1], 1
	towers[v] = t
res = {v: [] for v in variants}
with torch.no_grad():
	ref = None
	for v, t in towers.items():
		for _ in range(4):
			o = t(x)
		ref = o if ref is None else ref
		print(v, "max |d| vs one lane:", float((o - ref).abs().max()), flush=True)
	torch.cuda.synchronize()
	for rnd in range(5):
		for v, t in towers.items():
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(5):
				t(x)
			torch.cuda.synchronize()
			res[v].append((time.perf_counter() - t0) / 5)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}")
for v in variants:
	dt = statistics.median(res[v])
	print(f"lanes {v[0]} x {v[1]} CUs: {dt * 1e3:.3f} ms, {B / dt:.0f} img/s ({B / dt * cfg.flops_per_image() / 2.5e15:.3f})", flush=True)
