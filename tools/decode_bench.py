#!/usr/bin/env python3
"""Decode-only timing (greedy + beam-4, B=256, G=11 forced) for profiling: python tools/decode_bench.py [reps]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_decoder  # noqa: E402
from oracle import decoder_oracle as O  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spec = O.DecoderSpec(embed_dim=512, vocab_size=6912, token_length=12)
torch.manual_seed(1)
model, _ = make_decoder(spec, seed=None, device="cuda")
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
embed = torch.nn.functional.normalize(torch.randn(B, 512), dim=-1).cuda()
for name, fn in (("greedy", lambda: model.generate(embed, False, True, 1.0, 0.0, None, None, False)),
                 ("beam4", lambda: model.generate_beam(embed, 4, 1.0, 0.0, None, False, 0.0, None, False))):
	with torch.no_grad():
		for _ in range(3):
			fn()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(reps):
			out = fn()
		torch.cuda.synchronize()
		dt = (time.perf_counter() - t0) / reps
	print(f"{name}: {dt * 1e3:.2f} ms per call, {B / dt:.0f} labels/s, steps {out[0].shape[-1]}", flush=True)
