#!/usr/bin/env python3
"""Where the from-host end-to-end figure loses against resident images: the pipelined ViT-B/32 tower alone and with greedy decode, fed with resident / pinned-host batches.
python tools/from_host_probe.py [batch]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import clip_vit, embedders  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
vit = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).to(dev)
g = torch.Generator().manual_seed(B)
res = [torch.randn(B, 3, 224, 224, generator=g).to(dev) for _ in range(4)]
pinned = [t.cpu().pin_memory() for t in res]
greedy = lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)


def rate(src, dec, reps=4):
	with torch.no_grad():
		for _ in range(2):
			for e in embedders.pipeline_image_batches(vit, src, dev):
				dec(e)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for e in embedders.pipeline_image_batches(vit, src * reps, dev):
			dec(e)
		torch.cuda.synchronize()
	return B * len(src) * reps / (time.perf_counter() - t0)


for name, dec in (("tower only", lambda e: None), ("tower + greedy", greedy)):
	print(f"batch {B}, {name}: resident {rate(res, dec) / 1e3:.1f} k/s | from pinned host {rate(pinned, dec) / 1e3:.1f} k/s", flush=True)
stager = embedders.image_stager(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(12):
	_, _, slot = stager.stage(pinned[i % 4])
	embedders.ImageStager.release(slot, stager.copy_stream)
stager.copy_stream.synchronize()
print(f"H2D alone: {12 * pinned[0].numel() * 4 / (time.perf_counter() - t0) / 1e9:.1f} GB/s = {12 * B / (time.perf_counter() - t0) / 1e3:.1f} k images/s")
