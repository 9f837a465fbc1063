#!/usr/bin/env python3
"""Per-workgroup timeline of the 128^2 GEMM kernel (run on the GPU box): python tools/gemm128_timeline.py M,N,K [kc|ks|wgrad[:splits]] [resid]
kc: B [N][K] (forward);  ks: B [K][N] (input gradient);  wgrad: A [K][M], B [K][N], split-K with the fp32 atomic epilogue (weight gradient; splits
default to the model's choice).  Stamps: start, first K-tile in LDS, K loop done, epilogue done."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops, _lib  # noqa: E402

m, n, k = (int(x) for x in sys.argv[1].split(","))
mode = sys.argv[2] if len(sys.argv) > 2 else "kc"
ks = mode == "ks"
wg = mode.startswith("wgrad")
resid = "resid" in sys.argv
tiles = ((m + 127) // 128) * ((n + 127) // 128)
splits = int(mode.split(":")[1]) if ":" in mode else max(1, min(max(1, 512 // tiles), max(1, k // 512)))
a = (torch.rand(k, m, device="cuda") * 2 - 1).to(torch.bfloat16) if wg else (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
b = (torch.rand(k, n, device="cuda") * 2 - 1).to(torch.bfloat16) if (ks or wg) else (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
out = torch.zeros(m, n, dtype=torch.float32 if (resid or wg) else torch.bfloat16, device="cuda")
kw = dict(kind=ops.EPI_RESID_F32, resid=torch.randn(m, n, device="cuda")) if resid else {}
if wg:
	kw = dict(a_kstrided=True, kind=ops.EPI_ATOMIC_F32, split_k=splits, ldc=n)
ops.gemm_tile_policy(0)
run = lambda: ops.gemm(a, b, m, n, k, out=out, b_kstrided=ks or wg, **kw)
for _ in range(3):
	run()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
	run()
e.record()
torch.cuda.synchronize()
nwg = ((m + 127) // 128) * ((n + 127) // 128) * (((splits + 7) // 8 * 8) if wg and splits > 1 else 1)
buf = torch.zeros(16384 * 4, dtype=torch.int64, device="cuda")
_lib.lib().novic_gemm128_trace(ctypes.c_void_p(buf.data_ptr()))
run()
torch.cuda.synchronize()
_lib.lib().novic_gemm128_trace(ctypes.c_void_p(0))
t = buf.cpu().view(16384, 4)[:min(nwg, 16384)].double() / 100.0
t = t[t[:, 3] > 0]  # workgroups with an empty K range return before their first stamp is followed by the others
t0 = t[:, 0].min()
print(f"{m}x{n}x{k} {mode}{' resid' if resid else ''}{f' splits {splits}' if wg else ''}: {s.elapsed_time(e) * 100:.1f} us per launch, {nwg} workgroups, span {float(t[:, 3].max() - t0):.1f} us")
print(f"mean per workgroup: prologue {float((t[:, 1] - t[:, 0]).mean()):.2f} us, K loop {float((t[:, 2] - t[:, 1]).mean()):.2f} us ({float((t[:, 2] - t[:, 1]).mean()) / max((k // (splits if wg else 1)) // 64, 1):.3f} per K-tile), "
      f"epilogue {float((t[:, 3] - t[:, 2]).mean()):.2f} us, total {float((t[:, 3] - t[:, 0]).mean()):.2f} us")
