#!/usr/bin/env python3
"""Per-workgroup timeline of the 128^2 GEMM kernel (run on the GPU box): python tools/gemm128_timeline.py M,N,K [kc|ks] [resid]
kc: B [N][K] (forward);  ks: B [K][N] (input gradient).  Stamps: start, first K-tile in LDS, K loop done, epilogue done."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops, _lib  # noqa: E402

m, n, k = (int(x) for x in sys.argv[1].split(","))
ks = len(sys.argv) > 2 and sys.argv[2] == "ks"
resid = "resid" in sys.argv
a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
b = (torch.rand(k, n, device="cuda") * 2 - 1).to(torch.bfloat16) if ks else (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
out = torch.empty(m, n, dtype=torch.float32 if resid else torch.bfloat16, device="cuda")
kw = dict(kind=ops.EPI_RESID_F32, resid=torch.randn(m, n, device="cuda")) if resid else {}
ops.gemm_tile_policy(0)
run = lambda: ops.gemm(a, b, m, n, k, out=out, b_kstrided=ks, **kw)
for _ in range(3):
	run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
	run()
e.record()
torch.cuda.synchronize()
nwg = ((m + 127) // 128) * ((n + 127) // 128)
buf = torch.zeros(16384 * 4, dtype=torch.int64, device="cuda")
_lib.lib().novic_gemm128_trace(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
_lib.lib().novic_gemm128_trace(ctypes.c_void_p(0))
t = buf.cpu().view(16384, 4)[:min(nwg, 16384)].double() / 100.0
t0 = t[:, 0].min()
print(f"{m}x{n}x{k} {'ks' if ks else 'kc'}{' resid' if resid else ''}: {s.elapsed_time(e) * 100:.1f} us per launch, {nwg} workgroups, span {float(t[:, 3].max() - t0):.1f} us")
print(f"mean per workgroup: prologue {float((t[:, 1] - t[:, 0]).mean()):.2f} us, K loop {float((t[:, 2] - t[:, 1]).mean()):.2f} us ({float((t[:, 2] - t[:, 1]).mean()) / max(k // 64, 1):.3f} per K-tile), "
      f"epilogue {float((t[:, 3] - t[:, 2]).mean()):.2f} us, total {float((t[:, 3] - t[:, 0]).mean()):.2f} us")
