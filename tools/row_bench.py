#!/usr/bin/env python3
"""Row-kernel micro-benchmark at the training shapes (run on the GPU box): LayerNorm forward / backward on [57344 x 512].
   python tools/row_bench.py [reps]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda:0")
rows, E = 57344, 512  # 8192 samples x 7 label positions
g = torch.Generator(device="cpu").manual_seed(0)
x = torch.randn(rows, E, generator=g).to(dev)
dy = torch.randn(rows, E, generator=g).to(dev).bfloat16()
gamma = torch.randn(E, generator=g).to(dev)
dx = torch.randn(rows, E, generator=g).to(dev)
gout = torch.empty(rows, E, device=dev, dtype=torch.bfloat16)
y = torch.empty(rows, E, device=dev, dtype=torch.bfloat16)
dgamma = torch.zeros(E, device=dev)


def timeit(fn, bytes_):
	for _ in range(5):
		fn()
	torch.cuda.synchronize()
	a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	a.record()
	for _ in range(reps):
		fn()
	b.record()
	torch.cuda.synchronize()
	us = a.elapsed_time(b) * 1e3 / reps
	return us, bytes_ / us / 1e6


drop = ops.Dropout(0.1, 1234, 7)
cases = {
	"ln_fwd": (lambda: ops.layernorm_fwd(x, gamma, y, rows, E), rows * E * 6),
	"ln_bwd dx_in+g_out+dropout": (lambda: ops.layernorm_bwd(dy, x, gamma, dx, dx, gout, dgamma, rows, E, dropout=drop), rows * E * (4 + 4 + 2 + 4 + 2)),
	"ln_bwd dx_in+g_out": (lambda: ops.layernorm_bwd(dy, x, gamma, dx, dx, gout, dgamma, rows, E), rows * E * (4 + 4 + 2 + 4 + 2)),
	"ln_bwd dx_in": (lambda: ops.layernorm_bwd(dy, x, gamma, dx, dx, None, dgamma, rows, E), rows * E * (4 + 4 + 2 + 4)),
	"ln_bwd no dgamma": (lambda: ops.layernorm_bwd(dy, x, gamma, dx, dx, gout, None, rows, E), rows * E * (4 + 4 + 2 + 4 + 2)),
	"ln_bwd first (no dx_in)": (lambda: ops.layernorm_bwd(dy, x, gamma, None, dx, gout, dgamma, rows, E), rows * E * (4 + 2 + 4 + 2)),
}
for name, (fn, nbytes) in cases.items():
	us, tbs = timeit(fn, nbytes)
	print(f"{name:32s} {us:8.1f} us  {tbs:5.2f} TB/s", flush=True)
