#!/usr/bin/env python3
"""ViT-B/32 + greedy / beam-4 end to end at a caller batch of 256, the tower COALESCED over n consecutive batches (embedders.pipeline_image_batches(coalesce = n)) and up to `rows` rows of a launch DECODED in one call, by source
(resident fp32, pinned host fp32, pinned host uint8) and by the tower's workgroup budget.  python tools/e2e_coalesce.py [budgets ...] (default: pipeline_budget)
$E2E_COALESCE = "4:1024,5:1280" restricts the (batches per launch : decode rows) pairs; $E2E_SOURCES = "resident" the sources."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import clip_vit, embedders  # noqa: E402
from novic_amd.infer import split_decode_groups  # noqa: E402

dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
vit = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).to(dev)
vit.half_stream = os.environ.get("E2E_HALF", "0") == "1"  # (round 6: the half-precision residual stream local_clip.OpenAIEmbedder runs)
B = 256
g = torch.Generator().manual_seed(B)
u8 = [torch.randint(0, 256, (B, 3, 224, 224), generator=g, dtype=torch.uint8).pin_memory() for _ in range(30)]  # (30: a multiple of 2, 3, 5, 6, 10 -- whole groups)
mean, std = (torch.tensor(v).view(1, 3, 1, 1) for v in vit._pixel_norm())
f32 = [((u.float() / 255.0 - mean) / std).pin_memory() for u in u8]
res = [x.to(dev) for x in f32]
greedy = lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)
beam4 = lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)


def rate(src, dec, n, cus, rows, reps=2):
	def run(batches):
		for e, sizes in embedders.pipeline_image_batches(vit, batches, dev, cus, coalesce=n, grouped=True):
			for (a, b), _ in split_decode_groups(sizes, rows):
				dec(e[a:b])
	with torch.no_grad():
		for _ in range(3):
			run(src)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		run(src * reps)
		torch.cuda.synchronize()
	return B * len(src) * reps / (time.perf_counter() - t0)


budgets = [int(a) for a in sys.argv[1:]] or [None]
pairs = ((1, 256), (2, 512), (3, 768), (4, 512), (4, 1024), (5, 1280), (6, 1536), (8, 1024), (8, 2048), (10, 2560))
if os.environ.get("E2E_COALESCE"):
	pairs = tuple(tuple(int(v) for v in p.split(":")) for p in os.environ["E2E_COALESCE"].split(","))
wanted = os.environ.get("E2E_SOURCES", "resident,host fp32,host uint8").split(",")
for cus in budgets:
	for n, rows in pairs:
		line = [f"budget {cus} coalesce {n} decode rows {rows}:"]
		for sname, src in ((a, b) for a, b in (("resident", res), ("host fp32", f32), ("host uint8", u8)) if a in wanted):
			if os.environ.get("E2E_GREEDY_ONLY"):  # (tools/trace_e2e.sh: one decoder per trace)
				line.append(f"{sname} greedy {rate(src, greedy, n, cus, rows) / 1e3:.1f} k")
				continue
			line.append(f"{sname} greedy {rate(src, greedy, n, cus, rows) / 1e3:.1f} k / beam-4 {rate(src, beam4, n, cus, rows) / 1e3:.1f} k")
		print(" | ".join(line), flush=True)
