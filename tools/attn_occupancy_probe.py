#!/usr/bin/env python3
"""dec_attn_fwd / dec_attn_bwd at the training step's shape (8192 sequences x 10 positions, packed to ~61.5 k rows, 8 heads of 64, dropout 0.1) under the library
$NOVIC_HIP_LIB names -- a build of attention.hip that asks for extra, unused LDS per workgroup (tools/attn_occupancy_probe.sh), i.e. the same kernels at fewer resident
waves per CU.  python tools/attn_occupancy_probe.py <extra KiB, for the label>"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from novic_amd import ops  # noqa: E402
from novic_amd.ops import Dropout  # noqa: E402

label = sys.argv[1] if len(sys.argv) > 1 else "0"
A, S, P, H, D = 8192, 10, 4, 8, 64
E = H * D
g = torch.Generator().manual_seed(1)
lens = P - 1 + torch.randint(2, 8, (A,), generator=g)  # kept positions per sequence: the bench's label lengths U{1..6} + END
lens = lens.clamp(max=S)
key_pad = (torch.arange(S).unsqueeze(0) >= lens.unsqueeze(1)).to(torch.uint8).cuda()
start, ln = torch.zeros(A, dtype=torch.int32, device="cuda"), torch.zeros(A, dtype=torch.int32, device="cuda")
total = torch.zeros(1 + (A + 1023) // 1024, dtype=torch.int32, device="cuda")
ops.seq_layout(key_pad, A, S, start, ln, total)
rows = int(lens.sum())
M = A * S
# several buffer sets, rotated: a launch finds its operands as cold as the step's launches do (what a kernel has just WRITTEN is cold; 6 x 440 MB > the Infinity Cache)
sets = [dict(qkv=(torch.randn(M, 3 * E, device="cuda") * 0.5).to(torch.bfloat16), do=torch.randn(M, E, device="cuda").to(torch.bfloat16),
             o=torch.empty(M, E, dtype=torch.bfloat16, device="cuda"), dqkv=torch.empty(M, 3 * E, dtype=torch.bfloat16, device="cuda")) for _ in range(4)]
drop = Dropout(0.1, 1234, 5)


def timed(fn, reps=12):
	for i in range(4):
		fn(sets[i % 4])
	torch.cuda.synchronize()
	s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	s.record()
	for i in range(reps):
		fn(sets[i % 4])
	e.record()
	torch.cuda.synchronize()
	return s.elapsed_time(e) / reps * 1000


fwd = min(timed(lambda b: ops.dec_attn_fwd(b["qkv"], key_pad, b["o"], A, S, H, D, P, False, drop, seq=(start, ln))) for _ in range(3))
bwd = min(timed(lambda b: ops.dec_attn_bwd(b["qkv"], key_pad, b["do"], b["dqkv"], A, S, H, D, P, False, drop, seq=(start, ln))) for _ in range(3))
fb, bb = rows * E * 2 * 4, rows * E * 2 * 7
print(f"extra LDS {label:>4s} KiB per workgroup: dec_attn_fwd {fwd:7.1f} us ({fb / fwd / 1e6:4.2f} TB/s)   dec_attn_bwd {bwd:7.1f} us ({bb / bwd / 1e6:4.2f} TB/s)   [{rows} packed rows]", flush=True)
if label == "0":  # the GEMM a fused prologue would have to do per row tile, on the two tile sizes that exist: out-projection dX [rows x 512 x 512] on 256 x 256 and on 128 x 128 tiles
	w = (torch.randn(E, E, device="cuda") * E ** -0.5).to(torch.bfloat16)
	lim = torch.tensor([rows], dtype=torch.int32, device="cuda")
	for pol, name in ((1, "256 x 256 tiles (the step's launch)"), (0, "128 x 128 tiles, weights through L2 per tile")):
		prev = ops.gemm_tile_policy(pol)
		t = min(timed(lambda b: ops.gemm(b["do"], w, M, E, E, out=b["o"], row_limit=lim)) for _ in range(3))
		ops.gemm_tile_policy(prev)
		print(f"out-projection dX [{rows} x {E} x {E}] on {name}: {t:7.1f} us ({rows * E * 4 / t / 1e6:4.2f} TB/s of its 2 x rows x E x 2 bytes)", flush=True)
if label == "0":
	# Locality experiment: the same problems (sequence x head pairs, bytes, arithmetic) with every head's Q | K | V rows CONTIGUOUS -- emulated by presenting the kernels a
	# one-head model over 8 x as many sequences ([rows x 8][3 x 64]: a tile's operands are 16 x 384 contiguous bytes instead of 128-byte pieces of 3 KiB rows).  What a
	# head-major QKV layout written by the QKV GEMM's epilogue could at most buy the attention kernels.
	A1 = A * H
	lens1 = lens.repeat_interleave(H)
	kp1 = (torch.arange(S).unsqueeze(0) >= lens1.unsqueeze(1)).to(torch.uint8).cuda()
	st1, ln1 = torch.zeros(A1, dtype=torch.int32, device="cuda"), torch.zeros(A1, dtype=torch.int32, device="cuda")
	tot1 = torch.zeros(1 + (A1 + 1023) // 1024, dtype=torch.int32, device="cuda")
	ops.seq_layout(kp1, A1, S, st1, ln1, tot1)
	view = lambda t, w: t.view(-1)[: M * H * w].view(M * H, w)  # (the same buffers, read as [rows x 8][w])
	f1 = min(timed(lambda b: ops.dec_attn_fwd(view(b["qkv"], 3 * D), kp1, view(b["o"], D), A1, S, 1, D, P, False, drop, seq=(st1, ln1))) for _ in range(3))
	b1 = min(timed(lambda b: ops.dec_attn_bwd(view(b["qkv"], 3 * D), kp1, view(b["do"], D), view(b["dqkv"], 3 * D), A1, S, 1, D, P, False, drop, seq=(st1, ln1))) for _ in range(3))
	print(f"head-contiguous operands (one-head model over {A1} sequences): dec_attn_fwd {f1:7.1f} us ({fb / f1 / 1e6:4.2f} TB/s)   dec_attn_bwd {b1:7.1f} us ({bb / b1 / 1e6:4.2f} TB/s)", flush=True)
