#!/usr/bin/env python3
"""The prefix MLP's weight gradient dW[2048 x 512] = dprefix^T embn over the step's 8192 embeddings: the 128^2 split-K kernel with fp32 atomics (44.5 us in the step,
0.15 of the MFMA peak) against the 256-wide weight-gradient kernel with 4 / 8 / 16 parts (wgrad_supported() keeps it off that kernel: K < 16384).  python tools/prefix_dw_ab.py"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402
from novic_amd.embedding_decoder import _splits_for  # noqa: E402

M, N, K = 2048, 512, 8192
g = torch.Generator().manual_seed(1)
dy = (torch.randn(K, M, generator=g) * 0.3).to(torch.bfloat16).cuda()
x = (torch.randn(K, N, generator=g) * 0.3).to(torch.bfloat16).cuda()


def time_once(fn, n=20):
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for _ in range(n):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / n * 1000


tiles = ((M + 127) // 128) * ((N + 127) // 128)
out = torch.zeros(M, N, device="cuda")
variants = {"split-K atomics (128^2)": lambda: ops.gemm(dy, x, M, N, K, a_kstrided=True, b_kstrided=True, kind=ops.EPI_ATOMIC_F32, out=out, split_k=_splits_for(tiles, K), ldc=N)}
for s in (4, 8, 16):
	variants[f"wgrad256p, {s} parts"] = (lambda s_: (lambda: ops.wgrad(dy, x, M, N, K, out, splits=s_)))(s)
res = {k: [] for k in variants}
ref = dy.float().T @ x.float()
for k, fn in variants.items():
	out.zero_()
	fn()
	err = float((out - ref).abs().max() / ref.abs().max())
	print(f"{k}: relative error vs fp32 matmul {err:.2e}")
	for _ in range(3):
		fn()
torch.cuda.synchronize()
for rnd in range(7):
	for k, fn in variants.items():
		res[k].append(time_once(fn))
for k in variants:
	t = statistics.median(res[k])
	print(f"{k:28s}: {t:6.1f} us  {2.0 * M * N * K / t / 1e6:5.0f} TFLOP/s")
