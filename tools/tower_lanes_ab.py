#!/usr/bin/env python3
"""A tower with its batch on one stream against two sub-batches on streams of their own (NativeViT.lanes), in one process: python tools/tower_lanes_ab.py [CFG] [batch]
(run it under GPU_MAX_HW_QUEUES=4 and =8: with 4 hardware queues two lanes may share one and run one after the other)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import clip_vit  # noqa: E402

cfg = getattr(clip_vit, sys.argv[1] if len(sys.argv) > 1 else "VIT_L_14")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
vit = clip_vit.NativeViT(cfg, seed=3).cuda()
x = torch.randn(B, 3, cfg.image_size, cfg.image_size).cuda()
res = {1: [], 2: []}
with torch.no_grad():
	for rnd in range(4):
		for lanes in (1, 2):
			vit.lanes = lanes
			for _ in range(2):
				vit(x)
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(3):
				vit(x)
			torch.cuda.synchronize()
			res[lanes].append((time.perf_counter() - t0) / 3)
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}: " + " | ".join(f"{l} lane(s) {B / min(v):7.0f} img/s (best of 4; {min(v) * 1e3:.2f} ms)" for l, v in res.items()), flush=True)
