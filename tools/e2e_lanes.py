#!/usr/bin/env python3
"""ViT-B/32 + decode over a sequence of image batches: the tower pipelined against the decoding of the previous batch (embedders.pipeline_image_batches; one decode at a
time) against the same with the decoding of GROUPS of batches as concurrent lanes (generate_many / generate_beam_many): a decode step is ~30 dependent small launches
whatever its rows, so three batches decode in little more than the time of one, beside the towers of the next group.  python tools/e2e_lanes.py [batch] [host|resident]"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import clip_vit, embedders  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
host = len(sys.argv) > 2 and sys.argv[2] == "host"
dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
vit = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).to(dev)
g = torch.Generator().manual_seed(5)
NB = 12
seq = [torch.randn(B, 3, 224, 224, generator=g) for _ in range(4)]
seq = [(s.pin_memory() if host else s.to(dev)) for s in seq]
batches = [seq[i % 4] for i in range(NB)]


def run(lanes, cus, beam):
	def dec_many(es):
		if beam:
			return model.generate_beam_many(es, 4, 1.0, 0.0, None, False, 0.0, None, False) if len(es) > 1 else [model.generate_beam(es[0], 4, 1.0, 0.0, None, False, 0.0, None, False)]
		return model.generate_many(es, False, True, 1.0, 0.0, None, None, False) if len(es) > 1 else [model.generate(es[0], False, True, 1.0, 0.0, None, None, False)]
	group = []
	for e in embedders.pipeline_image_batches(vit, batches, dev, cus, ahead=lanes):
		group.append(e)
		if len(group) == lanes:
			dec_many(group)
			group = []
	if group:
		dec_many(group)


with torch.no_grad():
	for beam in (False, True):
		res = {}
		variants = [(1, 208), (2, 208), (3, 208), (3, 232), (3, 256), (4, 208)]
		for v in variants:
			for _ in range(2):
				run(v[0], v[1], beam)
			res[v] = []
		torch.cuda.synchronize()
		for rnd in range(4):
			for v in variants:
				torch.cuda.synchronize()
				t0 = time.perf_counter()
				run(v[0], v[1], beam)
				torch.cuda.synchronize()
				res[v].append((time.perf_counter() - t0) / NB)
		for v in variants:
			dt = statistics.median(res[v])
			print(f"{'beam-4' if beam else 'greedy'} {'host' if host else 'resident'} images, decode lanes {v[0]}, tower on {v[1]} CUs: {dt * 1e3:.3f} ms per batch, {B / dt:.0f} labels/s", flush=True)
