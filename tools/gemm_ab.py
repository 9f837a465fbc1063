#!/usr/bin/env python3
"""A/B of the two K-loop schedules of the 256 x 256 GEMM tile in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): gemm256_kernel (one barrier +
vmcnt(0) per K-tile) against gemm256p_kernel (8-phase schedule), on the shapes of the training step and the towers, uniform random operands in [-1, 1).
python tools/gemm_ab.py   (one MI355X)"""
import os
import statistics
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


def main():
	shapes = [("QKV", 61519, 1536, 512, "bf16"), ("logits", 36943, 6912, 512, "bf16"), ("in-proj dX", 61519, 512, 1536, "bf16"), ("logits dX", 36943, 512, 6912, "bf16"),
	          ("ViT-B/32 QKV", 12800, 2304, 768, "bias"), ("ViT-B/32 fc1", 12800, 3072, 768, "qgelu"), ("ViT-L/14 fc1", 65792, 4096, 1024, "gelu"),
	          ("ViT-L/14 fc2", 65792, 1024, 4096, "resid"), ("4096^3", 4096, 4096, 4096, "bf16"), ("8192^3", 8192, 8192, 8192, "bf16")]
	ops.gemm_tile_policy(2)
	for name, M, N, K, mode in shapes:
		a = (torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16)
		b = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
		kw = {}
		out = torch.empty(M, N, dtype=torch.float32 if mode == "resid" else torch.bfloat16, device="cuda")
		if mode == "resid":
			kw = dict(kind=ops.EPI_RESID_F32, resid=torch.randn(M, N, device="cuda"), bias=torch.randn(N, device="cuda"), split_tail=True)
		elif mode in ("bias", "qgelu", "gelu"):
			kw = dict(bias=torch.randn(N, device="cuda"), act={"bias": ops.ACT_NONE, "qgelu": ops.ACT_QUICKGELU, "gelu": ops.ACT_GELU}[mode])
		fn = lambda: ops.gemm(a, b, M, N, K, out=out, **kw)
		res = {0: [], 1: []}
		for pol in (0, 1):
			ops.gemm256_pipeline(pol)
			for _ in range(3):
				fn()
		torch.cuda.synchronize()
		for rnd in range(7):
			for pol in (0, 1):
				ops.gemm256_pipeline(pol)
				res[pol].append(time_once(fn))
		ops.gemm256_pipeline(1)
		fl = 2.0 * M * N * K
		m0, m1 = statistics.median(res[0]), statistics.median(res[1])
		print(f"{name:14s} [{M} x {N} x {K}] {mode:6s}: one-barrier {m0:8.1f} us (min {min(res[0]):8.1f}, {fl / m0 / 1e6:5.0f} TF) | 8-phase {m1:8.1f} us (min {min(res[1]):8.1f}, "
		      f"{fl / m1 / 1e6:5.0f} TF)  x{m0 / m1:.3f}", flush=True)


if __name__ == "__main__":
	main()
