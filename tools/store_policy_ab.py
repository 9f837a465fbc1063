#!/usr/bin/env python3
"""A/B of the output-store policy of the 256-wide GEMM tiles inside a tower: non-temporal (default; measured best for the training step's huge GEMMs, whose operand panels
must survive in L2) against ordinary write-back stores (the tower's activations -- hid 79 MB, the fp32 stream 39 MB at ViT-B/32 batch 256 -- could then be served to the
next GEMM out of L2 / the 256 MB Infinity Cache).  Each variant captures its own graphs.  python tools/store_policy_ab.py [CFG] [batch]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import _lib, clip_text, clip_vit  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "VIT_B_32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
if name.startswith("TEXT"):
	cfg = getattr(clip_text, name)
	mk = lambda: clip_text.NativeTextTower(cfg, seed=3).cuda()
	x = torch.randint(1, 49406, (B, 77)).cuda()
	fl = cfg.flops_per_text()
else:
	cfg = getattr(clip_vit, name)
	mk = lambda: clip_vit.NativeViT(cfg, seed=3).cuda()
	x = torch.randn(B, 3, cfg.image_size, cfg.image_size).cuda()
	fl = cfg.flops_per_image()
towers = {}
with torch.no_grad():
	for pol in (4, 5):
		_lib.lib().novic_gemm256_pipeline(pol)
		t = mk()
		for _ in range(4):
			o = t(x)
		towers[pol] = (t, o.clone())
	torch.cuda.synchronize()
	res = {4: [], 5: []}
	for rnd in range(7):
		for pol, (t, _) in towers.items():
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(5):
				t(x)
			torch.cuda.synchronize()
			res[pol].append((time.perf_counter() - t0) / 5)
_lib.lib().novic_gemm256_pipeline(4)
print("bit-identical outputs:", bool(torch.equal(towers[4][1], towers[5][1])))
for pol, label in ((4, "non-temporal"), (5, "write-back")):
	dt = statistics.median(res[pol])
	print(f"{name} batch {B} {label:12s}: {dt * 1e3:.3f} ms, {B / dt:.0f} /s ({B / dt * fl / 2.5e15:.3f} of the bf16 MFMA peak)", flush=True)
