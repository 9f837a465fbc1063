#!/usr/bin/env python3
"""Image batches through tower + decoder: one after the other on one stream against a two-stream pipeline (the tower of batch i + 1 beside the decode of batch i), the
tower's persistent GEMM grids on all 256 CUs or on fewer (ops.cu_budget: the rest stays free for the decode step's small kernels).
python tools/e2e_overlap.py [VIT_B_32|VIT_L_14]   (one MI355X)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from novic_amd import clip_vit, ops  # noqa: E402

dev = torch.device("cuda")
CFG = getattr(clip_vit, sys.argv[1] if len(sys.argv) > 1 else "VIT_B_32")
spec = bench.WorkloadSpec(embed_dim=CFG.embed_dim, vocab_size=6912, token_length=12)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
vit = clip_vit.NativeViT(CFG, seed=3).to(dev)
B, NB = 256, (12 if CFG is clip_vit.VIT_B_32 else 5)
g = torch.Generator().manual_seed(1)
batches = [torch.randn(B, 3, 224, 224, generator=g).to(dev) for _ in range(NB)]
for name, dec in (("greedy", lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)), ("beam4", lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False))):
	with torch.no_grad():
		for _ in range(3):
			dec(vit(batches[0]))
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for im in batches:
			dec(vit(im))
		torch.cuda.synchronize()
		serial = time.perf_counter() - t0
		# pipeline: tower on stream A, decode on the main stream
		sa = torch.cuda.Stream()
		main = torch.cuda.current_stream()
		def tower(im, cus):
			sa.wait_stream(main)
			with ops.cu_budget(cus), torch.cuda.stream(sa):
				e = vit(im)
			ev = torch.cuda.Event()
			ev.record(sa)
			return e, ev
		res = []
		for cus in (256, 248, 232, 208, 184, 160):
			for rep in range(2):  # (the first pass captures the tower's graph for this grid size)
				torch.cuda.synchronize()
				t0 = time.perf_counter()
				nxt = tower(batches[0], cus)
				for i in range(NB):
					e, ev = nxt
					if i + 1 < NB:
						nxt = tower(batches[i + 1], cus)
					main.wait_event(ev)
					e.record_stream(main)
					dec(e)
				torch.cuda.synchronize()
				piped = time.perf_counter() - t0
			res.append(f"{cus} CUs {NB * B / piped:7.0f} ({piped / NB * 1e3:.2f} ms)")
	print(f"{name}: serial {NB * B / serial:8.0f} labels/s ({serial / NB * 1e3:.2f} ms per batch) | tower of the next batch beside the decode, tower GEMM grids on " + ", ".join(res), flush=True)
