#!/usr/bin/env python3
"""The text tower's out-projection + residual [19 712 x 512 x 512] on the streaming 128-column kernel (tile policy 1: what the decoder's training shapes get) against the 256 x 256
tile (policy 2), inside the tower (one tower captured per setting) and isolated.  python tools/text_proj_ab.py [batch]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import clip_text, ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randint(1, 49000, (B, 77)).cuda()
x[:, -1] = 49407
with torch.no_grad():
	towers, outs = {}, {}
	for pol in (1, 2):
		ops.gemm_tile_policy(pol)
		towers[pol] = clip_text.NativeTextTower(clip_text.TEXT_B_32, seed=3).cuda()
		for _ in range(4):
			outs[pol] = towers[pol](x)
	torch.cuda.synchronize()
	res = {1: [], 2: []}
	for rnd in range(7):
		for pol, t in towers.items():
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(5):
				t(x)
			torch.cuda.synchronize()
			res[pol].append((time.perf_counter() - t0) / 5)
ops.gemm_tile_policy(1)
print(f"text tower batch {B}: " + " | ".join(f"policy {p}: {statistics.median(v) * 1e3:.3f} ms, {B / statistics.median(v):.0f} texts/s" for p, v in res.items())
      + f" | bit-identical: {bool(torch.equal(outs[1], outs[2]))}, max |d| {float((outs[1] - outs[2]).abs().max()):.2e}")
