#!/usr/bin/env python3
"""Why a tower's GEMMs run slower inside the forward than in a loop of their own (ViT-B/32, batch 256: proj 31 -> 42 us, fc1 62 -> 75, fc2 66 -> 86, qkv 44 -> 51):
each shape timed (a) on the same operands every launch (what tools/vit_b32_gemm_ab.py does: weights in L2, activations in the Infinity Cache), (b) on operands rotated over
enough buffer sets that none is cached, (c) like (b) with the activation operand rewritten just before the launch by a copy kernel (what the forward does: the operand was
just produced), (d) like (a) behind an unrelated memory-bound kernel (the clocks / caches a GEMM meets after an attention or LayerNorm launch).  python tools/gemm_cold_ab.py [rows] [width]"""
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402

M = int(sys.argv[1]) if len(sys.argv) > 1 else 12800
W = int(sys.argv[2]) if len(sys.argv) > 2 else 768
SETS = 12
dev = "cuda"
junk_a, junk_b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)  # 256 MiB each


def timed(fn, pre=None, n=24):
	"""median launch time, the launches enqueued back to back (no host synchronisation in between: the GPU never idles), one event pair per launch"""
	evs = []
	for i in range(n):
		if pre is not None:
			pre(i)
		s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		s.record()
		fn(i)
		e.record()
		evs.append((s, e))
	torch.cuda.synchronize()
	return statistics.median([s.elapsed_time(e) * 1000 for s, e in evs[4:]])


def looped(fn, n=24):
	"""the same without the inner events: n launches between ONE event pair"""
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for i in range(n):
		fn(i)
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) * 1000 / n


for name, N, K, mode in (("proj", W, W, "resid"), ("fc2", W, 4 * W, "resid"), ("qkv", 3 * W, W, "bias"), ("fc1", 4 * W, W, "qgelu")):
	A = [(torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16) for _ in range(SETS)]
	B = [(torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16) for _ in range(SETS)]
	src = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
	bias = torch.randn(N, device=dev)
	if mode == "resid":
		O = [torch.empty(M, N, device=dev) for _ in range(SETS)]
		R = [torch.randn(M, N, device=dev) for _ in range(SETS)]
		kw = lambda i: dict(kind=ops.EPI_RESID_F32, resid=R[i], bias=bias, split_tail=True)
	else:
		O = [torch.empty(M, N, dtype=torch.bfloat16, device=dev) for _ in range(SETS)]
		kw = lambda i: dict(bias=bias, act=ops.ACT_QUICKGELU if mode == "qgelu" else ops.ACT_NONE, split_tail=True)
	run = lambda i: ops.gemm(A[i], B[i], M, N, K, out=O[i], **kw(i))
	for i in range(SETS):
		run(i)
	loop_hot, loop_cold = looped(lambda i: run(0)), looped(lambda i: run(i % SETS))
	hot = timed(lambda i: run(0))
	cold = timed(lambda i: run(i % SETS), pre=lambda i: junk_a.copy_(junk_b))
	fresh = timed(lambda i: run(i % SETS), pre=lambda i: (junk_a.copy_(junk_b), A[i % SETS].copy_(src)))
	fresh_w = timed(lambda i: run(i % SETS), pre=lambda i: (junk_a.copy_(junk_b), A[i % SETS].copy_(src), B[i % SETS].mul_(1.0)))
	behind = timed(lambda i: run(0), pre=lambda i: junk_a[: 8 << 20].copy_(junk_b[: 8 << 20]))
	print(f"{name:5s} [{M} x {N} x {K}]: loop of 24, one event pair: same operands {loop_hot:6.1f} us, rotating {loop_cold:6.1f} | event pair per launch: same operands {hot:6.1f} us | all cold {cold:6.1f} | activation just written {fresh:6.1f} | + weights just touched {fresh_w:6.1f} | "
	      f"same operands behind a 32 MiB copy {behind:6.1f}", flush=True)
