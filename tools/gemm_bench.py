#!/usr/bin/env python3
"""Micro-benchmark of novic_gemm_bf16 on the shapes of one decoder training step (run on the GPU box)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402

M, R, E, K, V = 81920, 57344, 512, 128, 6912
SHAPES = [  # name, (a_shape, b_shape), M, N, K, a_ks, b_ks, kind, split
	("fwd qkv      ", (M, E), (3 * E, E), M, 3 * E, E, 0, 0, ops.EPI_STORE_BF16, 1),
	("fwd out+resid", (M, E), (E, E), M, E, E, 0, 0, ops.EPI_RESID_F32, 1),
	("fwd w1+gelu  ", (M, E), (K, E), M, K, E, 0, 0, ops.EPI_GELU_BF16, 1),
	("fwd w2+resid ", (M, K), (E, K), M, E, K, 0, 0, ops.EPI_RESID_F32, 1),
	("fwd logits   ", (R, E), (V, E), R, V, E, 0, 0, ops.EPI_STORE_BF16, 1),
	("bwd dln1     ", (M, 3 * E), (3 * E, E), M, E, 3 * E, 0, 1, ops.EPI_STORE_BF16, 1),
	("bwd datt     ", (M, E), (E, E), M, E, E, 0, 1, ops.EPI_STORE_BF16, 1),
	("bwd dln2     ", (M, K), (K, E), M, E, K, 0, 1, ops.EPI_STORE_BF16, 1),
	("bwd dxf      ", (R, V), (V, E), R, E, V, 0, 1, ops.EPI_STORE_BF16, 1),
	("bwd dWqkv    ", (M, 3 * E), (M, E), 3 * E, E, M, 1, 1, ops.EPI_ATOMIC_F32, 10),
	("bwd dWo      ", (M, E), (M, E), E, E, M, 1, 1, ops.EPI_ATOMIC_F32, 32),
	("bwd dW1      ", (M, K), (M, E), K, E, M, 1, 1, ops.EPI_ATOMIC_F32, 128),
	("bwd dWtok    ", (R, V), (R, E), V, E, R, 1, 1, ops.EPI_ATOMIC_F32, 2),
]


def main():
	dev = "cuda"
	tot_t = tot_f = 0.0
	for name, ash, bsh, m, n, k, aks, bks, kind, split in SHAPES:
		a = (torch.randn(*ash, device=dev) * 0.3).to(torch.bfloat16)
		b = (torch.randn(*bsh, device=dev) * 0.3).to(torch.bfloat16)
		out_dtype = torch.float32 if kind in (ops.EPI_ATOMIC_F32, ops.EPI_RESID_F32, ops.EPI_STORE_F32) else torch.bfloat16
		out = torch.zeros(m, n, dtype=out_dtype, device=dev)
		kw = {}
		if kind == ops.EPI_RESID_F32:
			kw["resid"] = torch.randn(m, n, device=dev)
		if kind == ops.EPI_GELU_BF16:
			kw["out2"] = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
		run = lambda: ops.gemm(a, b, m, n, k, a_kstrided=bool(aks), b_kstrided=bool(bks), kind=kind, out=out, split_k=split, **kw)
		for _ in range(3):
			run()
		s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		torch.cuda.synchronize()
		s.record()
		reps = 10
		for _ in range(reps):
			run()
		e.record()
		torch.cuda.synchronize()
		us = s.elapsed_time(e) / reps * 1000
		fl = 2.0 * m * n * k
		tot_t += us
		tot_f += fl
		print(f"{name} M={m:6d} N={n:5d} K={k:6d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)
	print(f"sum: {tot_t:.0f} us, {tot_f / tot_t / 1e6:.1f} TFLOP/s (one of each)")


if __name__ == "__main__":
	main()
