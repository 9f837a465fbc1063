#!/usr/bin/env python3
"""Pipelined ViT-B/32 (or $E2E_TOWER = VIT_L_14 ...; $E2E_HALF = 1: half-precision residual stream) + greedy / beam-4 at a batch size over the tower's workgroup budget: python tools/e2e_budget_sweep.py [batch] [budgets ...]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import clip_vit, embedders  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
budgets = [int(a) for a in sys.argv[2:]] or [256, 232, 208, 184, 160]
dev = torch.device("cuda")
vit = clip_vit.NativeViT(getattr(clip_vit, os.environ.get("E2E_TOWER", "VIT_B_32")), seed=3).to(dev)
vit.half_stream = os.environ.get("E2E_HALF", "0") == "1"  # (round 6: the residual stream in IEEE half -- what local_clip.OpenAIEmbedder runs)
spec = bench.WorkloadSpec(embed_dim=vit.cfg.embed_dim, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
g = torch.Generator().manual_seed(B)
seq = [torch.randn(B, 3, 224, 224, generator=g).to(dev) for _ in range(3)]
for name, dec in (("greedy", lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)), ("beam-4", lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False))):
	out = []
	for cus in budgets:
		with torch.no_grad():
			for _ in range(2):
				for e in embedders.pipeline_image_batches(vit, seq, dev, cus):
					dec(e)
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for e in embedders.pipeline_image_batches(vit, seq * 3, dev, cus):
				dec(e)
			torch.cuda.synchronize()
		out.append(f"{cus} CUs {B * 9 / (time.perf_counter() - t0) / 1e3:.1f} k")
	print(f"batch {B} {name}: " + " | ".join(out), flush=True)
