#!/usr/bin/env python3
"""Per-workgroup timeline of the LDS-DMA GEMM kernel (run on the GPU box): python tools/gemm_timeline.py [M,N,K] [resid|bias|qgelu]
Stamps (100 MHz wall clock) per workgroup and tile: t0 tile start, t1 first K-tile done, t2 K loop done, t3 stores issued."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops, _lib  # noqa: E402

m, n, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "57344,6912,512").split(","))
a = (torch.rand(m, k, device="cuda") * 2 - 1).to(torch.bfloat16)
b = (torch.rand(n, k, device="cuda") * 2 - 1).to(torch.bfloat16)
mode = sys.argv[2] if len(sys.argv) > 2 else ""
resid_mode = mode == "resid"  # fp32 residual epilogue (ViT proj / fc2) instead of the bf16 store; "bias" / "qgelu": the bf16 store with a bias (and QuickGELU: ViT fc1)
out = torch.empty(m, n, dtype=torch.float32 if resid_mode else torch.bfloat16, device="cuda")
res = torch.randn(m, n, device="cuda") if resid_mode else None
kw = dict(kind=ops.EPI_RESID_F32, resid=res, bias=torch.randn(n, device="cuda")) if resid_mode else {}
if mode in ("bias", "qgelu"):
	kw = dict(bias=torch.randn(n, device="cuda"), act=ops.ACT_QUICKGELU if mode == "qgelu" else ops.ACT_NONE)
ops.gemm_tile_policy(2)
for _ in range(3):
	ops.gemm(a, b, m, n, k, out=out, **kw)
buf = torch.zeros(256 * 32 * 4, dtype=torch.int64, device="cuda")
_lib.lib().novic_gemm256_trace(ctypes.c_void_p(buf.data_ptr()))
ops.gemm(a, b, m, n, k, out=out, **kw)
torch.cuda.synchronize()
_lib.lib().novic_gemm256_trace(ctypes.c_void_p(0))
t = buf.cpu().view(256, 32, 4).double() / 100.0  # us
t0 = t[:, 0, 0].min()
ntile = int((t[0, :, 0] > 0).sum())
print(f"shape {m}x{n}x{k}: {ntile} traced tiles per workgroup; first tile starts spread over {float(t[:, 0, 0].max() - t0):.2f} us")
first = t[:, :ntile, 1] - t[:, :ntile, 0]
kloop = t[:, :ntile, 2] - t[:, :ntile, 0]
store = t[:, :ntile, 3] - t[:, :ntile, 2]
nxt = t[:, 1:ntile, 0] - t[:, :ntile - 1, 3]
print(f"per tile (mean over workgroups and tiles 1..): first K-tile {float(first[:, 1:].mean()):.2f} us, whole K loop {float(kloop[:, 1:].mean()):.2f} us, "
      f"store issue {float(store[:, 1:].mean()):.2f} us, gap to next tile {float(nxt.mean()):.2f} us")
print(f"tile 0 (cold): first K-tile {float(first[:, 0].mean()):.2f} us, whole K loop {float(kloop[:, 0].mean()):.2f} us, store issue {float(store[:, 0].mean()):.2f} us")
print(f"K-tiles after the first: {float((kloop[:, 1:] - first[:, 1:]).mean() / max(k // 64 - 1, 1)):.3f} us each")
for wg in (0, 1, 8, 100, 255):
	row = " ".join(f"{float(t[wg, i, 0] - t0):7.1f}/{float(first[wg, i]):4.1f}/{float(kloop[wg, i]):5.1f}/{float(store[wg, i]):4.1f}" for i in range(min(ntile, 6)))
	print(f"wg {wg:3d} start/first/kloop/store: {row}")
print(f"spread of tile-5 start across workgroups: {float(t[:, min(5, ntile - 1), 0].max() - t[:, min(5, ntile - 1), 0].min()):.2f} us; kernel span {float(t[:, :ntile, 3].max() - t0):.1f} us")
