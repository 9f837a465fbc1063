#!/usr/bin/env python3
"""Weight-gradient GEMMs of the training step in isolation: the 128^2 split-K atomic kernel (gemm.hip) against the 256-wide partial-sum kernel (wgrad.hip).
python tools/wgrad_bench.py   (one MI355X)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402
from novic_amd.embedding_decoder import _splits_for  # noqa: E402


def timeit(fn, n=20):
	for _ in range(3):
		fn()
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	torch.cuda.synchronize()
	s.record()
	for _ in range(n):
		fn()
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / n * 1000


def main():
	g = torch.Generator().manual_seed(0)
	for name, M, N, K in (("in-proj dW", 1536, 512, 61500), ("logits dW", 6912, 512, 36943), ("out-proj dW", 512, 512, 61500), ("prefix dW", 2048, 512, 8192),
	                      ("linear1 dW", 128, 512, 61500), ("linear2 dW", 512, 128, 61500)):
		dy = (torch.randn(K, M, generator=g) * 0.3).to(torch.bfloat16).cuda()
		x = (torch.randn(K, N, generator=g) * 0.3).to(torch.bfloat16).cuda()
		out = torch.zeros(M, N, device="cuda")
		tiles = ((M + 127) // 128) * ((N + 127) // 128)
		t_old = timeit(lambda: ops.gemm(dy, x, M, N, K, a_kstrided=True, b_kstrided=True, kind=ops.EPI_ATOMIC_F32, out=out, split_k=_splits_for(tiles, K), ldc=N))
		line = f"{name:12s} [{M} x {N}] over {K} rows: 128^2 split-K atomics {t_old:7.1f} us ({2 * M * N * K / t_old / 1e6:6.0f} TFLOP/s)"
		for S in (0, 8, 16, 32, 64):
			t256 = ((M + 255) // 256) * ((N + 255) // 256) if min(M, N) > 128 else (max(M, N) + 255) // 256
			if S and (t256 * S > 256 or (min(M, N) <= 128 and S < 32)):
				continue
			t_new = timeit(lambda: ops.wgrad(dy, x, M, N, K, out, splits=S))
			line += f" | 256-wide S={S or 256 // t256}: {t_new:7.1f} us ({2 * M * N * K / t_new / 1e6:6.0f})"
		print(line, flush=True)


if __name__ == "__main__":
	main()
