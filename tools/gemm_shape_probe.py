#!/usr/bin/env python3
"""One GEMM shape family at several row counts, operands cached (same buffers every launch) and cold (rotating over enough buffer sets): is a shape bound by what it pulls from
HBM?  python tools/gemm_shape_probe.py N K [rows ...]   (default: the in-projection input gradient, N = 512, K = 1536)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
rows = [int(a) for a in sys.argv[3:]] or [16384, 32768, 65536]
junk_a, junk_b = torch.empty(64 << 20, device="cuda"), torch.empty(64 << 20, device="cuda")


def loop(fn, n):
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for i in range(n):
		fn(i)
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) * 1000 / n


for M in rows:
	sets = max(2, int((600 << 20) / (M * (K + N) * 2)) + 1)
	A = [(torch.rand(M, K, device="cuda") * 2 - 1).to(torch.bfloat16) for _ in range(sets)]
	B = (torch.rand(N, K, device="cuda") * 2 - 1).to(torch.bfloat16)
	O = [torch.empty(M, N, dtype=torch.bfloat16, device="cuda") for _ in range(sets)]
	run = lambda i: ops.gemm(A[i % sets], B, M, N, K, out=O[i % sets])
	for i in range(sets):
		run(i)
	hot = loop(lambda i: run(0), 20)
	cold = loop(run, 3 * sets)
	fl = 2.0 * M * N * K
	print(f"[{M} x {N} x {K}] tile {ops.gemm_last_tile()}: operands cached {hot:7.1f} us {fl / hot / 1e6:6.0f} TFLOP/s | cold ({sets} buffer sets) {cold:7.1f} us {fl / cold / 1e6:6.0f} TFLOP/s "
	      f"({(M * (K + N) * 2) / cold / 1e6:.2f} TB/s of operand + output bytes)", flush=True)
	del A, O
