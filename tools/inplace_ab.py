#!/usr/bin/env python3
"""A/B of the in-place fp32 residual stream of the image tower (NativeViT.inplace_residual: proj / fc2 write the buffer they read the residual from) against two buffers in turn,
interleaved rounds in one process, graph replay.  python tools/inplace_ab.py [VIT_B_32|VIT_L_14|VIT_H_14] [batch] [attribute]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import clip_vit  # noqa: E402

ATTR = sys.argv[3] if len(sys.argv) > 3 else "inplace_residual"  # (any boolean class attribute of NativeViT: share_buffers, ...)
cfg = getattr(clip_vit, sys.argv[1] if len(sys.argv) > 1 else "VIT_B_32")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
x = torch.randn(B, 3, cfg.image_size, cfg.image_size).cuda()
towers = {}
for fold in (False, True):
	t = clip_vit.NativeViT(cfg, seed=3).cuda()
	setattr(t, ATTR, fold)
	towers[fold] = t
res = {False: [], True: []}
with torch.no_grad():
	outs = {}
	for fold, t in towers.items():
		for _ in range(4):
			outs[fold] = t(x)
	torch.cuda.synchronize()
	for rnd in range(7):
		for fold, t in towers.items():
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(5):
				t(x)
			torch.cuda.synchronize()
			res[fold].append((time.perf_counter() - t0) / 5)
cos = float((outs[False] * outs[True]).sum(dim=1).min())
for fold in (False, True):
	dt = statistics.median(res[fold])
	print(f"{sys.argv[1] if len(sys.argv) > 1 else 'VIT_B_32'} batch {B} {ATTR}={fold}: {dt * 1e3:.3f} ms, {B / dt:.0f} img/s, {B / dt * cfg.flops_per_image() / 2.5e15:.3f} of the bf16 MFMA peak", flush=True)
print(f"min cosine between the two forms' embeddings: {cos:.6f}; bit-identical: {bool(torch.equal(outs[False], outs[True]))}")
