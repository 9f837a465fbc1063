#!/usr/bin/env python3
"""A/B of the two weight-gradient kernels in ONE process, interleaved rounds (cdna_hip_programming.md rule 24): wgrad256_kernel (one barrier per K-tile) against
wgrad256p_kernel (8-phase schedule), on the shapes of the training step, random operands.  Prints median / min microseconds per launch pair (kernel + reduction).
python tools/wgrad_ab.py   (one MI355X)"""
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
	g = torch.Generator().manual_seed(0)
	K = 61519
	shapes = [("in-proj dW", 1536, 512, K), ("out-proj dW", 512, 512, K), ("logits dW", 6912, 512, 36943), ("linear1 dW", 128, 512, K), ("linear2 dW", 512, 128, K)]
	for name, M, N, Kk in shapes:
		dy = (torch.randn(Kk, M, generator=g) * 0.3).to(torch.bfloat16).cuda()
		x = (torch.randn(Kk, N, generator=g) * 0.3).to(torch.bfloat16).cuda()
		out = torch.zeros(M, N, device="cuda")
		fn = lambda: ops.wgrad(dy, x, M, N, Kk, out)
		res = {0: [], 1: []}
		for pol in (0, 1):
			ops.wgrad_policy(pol)
			for _ in range(3):
				fn()
		torch.cuda.synchronize()
		for rnd in range(8):
			for pol in (0, 1):
				ops.wgrad_policy(pol)
				res[pol].append(time_once(fn))
		ops.wgrad_policy(1)
		fl = 2.0 * M * N * Kk
		print(f"{name:12s} [{M} x {N}] K={Kk}: one-barrier median {statistics.median(res[0]):7.1f} min {min(res[0]):7.1f} us ({fl / statistics.median(res[0]) / 1e6:5.0f} TF) | "
		      f"8-phase median {statistics.median(res[1]):7.1f} min {min(res[1]):7.1f} us ({fl / statistics.median(res[1]) / 1e6:5.0f} TF)", flush=True)
	# the attention pair as the step launches it
	dy1 = (torch.randn(K, 1536, generator=g) * 0.3).to(torch.bfloat16).cuda(); x1 = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda()
	dy2 = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda(); x2 = (torch.randn(K, 512, generator=g) * 0.3).to(torch.bfloat16).cuda()
	o1, o2 = torch.zeros(1536, 512, device="cuda"), torch.zeros(512, 512, device="cuda")
	fn = lambda: ops.wgrad2(dy1, x1, 1536, 512, o1, dy2, x2, 512, 512, o2, K)
	res = {0: [], 1: []}
	for rnd in range(9):
		for pol in (0, 1):
			ops.wgrad_policy(pol)
			t = time_once(fn)
			if rnd:
				res[pol].append(t)
	ops.wgrad_policy(1)
	fl = 2.0 * 2048 * 512 * K
	print(f"attention pair [1536+512 x 512] K={K}: one-barrier median {statistics.median(res[0]):7.1f} us ({fl / statistics.median(res[0]) / 1e6:5.0f} TF) | "
	      f"8-phase median {statistics.median(res[1]):7.1f} us ({fl / statistics.median(res[1]) / 1e6:5.0f} TF)")


if __name__ == "__main__":
	main()
