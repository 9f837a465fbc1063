#!/usr/bin/env python3
"""Per-phase shader-clock timeline of ONE steady K-tile of the 8-phase GEMM (needs a library built with -DGEMM256_DIAG_PHASES=1, loaded through $NOVIC_HIP_LIB):
python tools/gemm_phases.py [M,N,K]   -- per wave group and phase: cycles in the LOAD segment, waiting at the opening barrier (+ lgkmcnt), in the MFMA block, at the closing barrier."""
import ctypes
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops, _lib  # noqa: E402

m, n, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,1536,512").split(","))
a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
out = torch.empty(m, n, dtype=torch.bfloat16, device="cuda")
ops.gemm_tile_policy(2)
for _ in range(3):
	ops.gemm(a, b, m, n, k, out=out)
buf = torch.zeros(256 * 32 * 4, dtype=torch.int64, device="cuda")
_lib.lib().novic_gemm256_trace(ctypes.c_void_p(buf.data_ptr()))
ops.gemm(a, b, m, n, k, out=out)
torch.cuda.synchronize()
_lib.lib().novic_gemm256_trace(ctypes.c_void_p(0))
t = buf.view(256, 32, 4).cpu()
rows = {0: [], 1: []}
for wg in range(256):
	st = t[wg, 16:24].reshape(2, 16)
	for grp in (0, 1):
		v = st[grp].tolist()
		if v[0] and all(x > 0 for x in v):
			rows[grp].append(v)
if not rows[0]:
	sys.exit("no phase stamps: is the library built with -DGEMM256_DIAG_PHASES=1 (NOVIC_HIP_LIB)?")
for grp in (0, 1):
	print(f"wave group {grp} (waves {4 * grp}-{4 * grp + 3}), median over {len(rows[grp])} workgroups, shader-clock cycles:")
	tot = 0
	for ph in range(4):
		seg = lambda i0, i1: statistics.median(r[i1] - r[i0] for r in rows[grp])
		load, wait, mma = seg(4 * ph, 4 * ph + 1), seg(4 * ph + 1, 4 * ph + 2), seg(4 * ph + 2, 4 * ph + 3)
		close = seg(4 * ph + 3, 4 * ph + 4) if ph < 3 else float("nan")
		tot += load + wait + mma + (close if ph < 3 else 0)
		print(f"  phase {ph}: LOAD {load:6.0f} | opening barrier + lgkmcnt {wait:6.0f} | 16 MFMAs {mma:6.0f} | closing barrier -> next LOAD {close:6.0f}")
	print(f"  phases 0-3 without the last closing barrier: {tot:.0f} cycles")
# the two groups against each other: group 1 runs one barrier behind
d = statistics.median(r1[0] - r0[0] for r0, r1 in zip(rows[0], rows[1]))
print(f"group 1 starts its phase 0 {d:.0f} cycles after group 0")
