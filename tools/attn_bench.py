#!/usr/bin/env python3
"""CLIP tower attention alone (novic_clip_attn_fwd): streaming kernel (policy 0) against the K/V-resident / blocked kernels (policy 1) at the released towers' shapes.
python tools/attn_bench.py   (one MI355X)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402


def time_once(fn, n=10):
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for _ in range(n):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / n * 1000


for name, B, N, H, D in (("ViT-H/14-378", 128, 730, 16, 80), ("same, head_dim 64", 128, 730, 16, 64), ("ViT-L/14", 256, 257, 16, 64), ("SigLIP B/16", 256, 196, 12, 64),
                         ("SO400M/14 (72 -> 80)", 256, 256, 16, 80), ("1024 tokens", 64, 1024, 16, 64)):
	W = H * D
	qkv = (torch.randn(B * N, 3 * W, device="cuda") * 1.0).to(torch.bfloat16)
	o = torch.empty(B * N, W, dtype=torch.bfloat16, device="cuda")
	res = []
	for pol in (0, 1):
		prev = ops.vit_attn_policy(pol)
		fn = lambda: ops.clip_attn_fwd(qkv, o, B, N, H, D, causal=False)
		for _ in range(3):
			fn()
		torch.cuda.synchronize()
		t = min(time_once(fn) for _ in range(3))
		ops.vit_attn_policy(prev)
		fl = 4.0 * B * H * N * N * D
		by = B * N * 4 * W * 2
		res.append(f"policy {pol}: {t:8.1f} us {fl / t / 1e6:6.0f} TFLOP/s {by / t / 1e6:5.2f} TB/s")
	print(f"{name:22s} [{B} x {N} tokens, {H} heads of {D}]  " + " | ".join(res), flush=True)
