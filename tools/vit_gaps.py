#!/usr/bin/env python3
"""Where a ViT-B/32 forward's wall time goes: eager against hipGraph replay, and -- from a rocprofv3 kernel trace of this script (tools/trace_gaps.sh) -- the gaps between
consecutive kernels.  python tools/vit_gaps.py [CFG] [batch]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import clip_vit  # noqa: E402

cfg = getattr(clip_vit, sys.argv[1] if len(sys.argv) > 1 else "VIT_B_32")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
vit = clip_vit.NativeViT(cfg, seed=3).cuda()
x = torch.randn(B, 3, cfg.image_size, cfg.image_size).cuda()
for graphs in (False, True):
	vit.use_graphs = graphs
	with torch.no_grad():
		for _ in range(4):
			vit(x)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(10):
			vit(x)
		torch.cuda.synchronize()
		dt = (time.perf_counter() - t0) / 10
		# host time to ENQUEUE one forward (no sync)
		torch.cuda.synchronize()
		t1 = time.perf_counter()
		vit(x)
		enq = time.perf_counter() - t1
		torch.cuda.synchronize()
	print(f"{'graph replay' if graphs else 'eager       '}: {dt * 1e3:.3f} ms per {B} images ({B / dt:.0f} img/s); host time to enqueue one forward {enq * 1e3:.3f} ms", flush=True)
