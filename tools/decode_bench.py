#!/usr/bin/env python3
"""Decode-only timing (greedy / beam-4 unguided with 11 forced steps, guided beam-10 over 37.6k synthetic nouns) for profiling:
python tools/decode_bench.py [reps] [batch] [greedy,beam4,beam10g] [lanes,...]   (lanes: that many independent batches at the same time, generate_many / generate_beam_many)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (product-side model builder; nothing under oracle/ or tests/ is imported here)

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
spec = bench.WorkloadSpec(embed_dim=512, vocab_size=6912, token_length=12)
torch.manual_seed(1)
model = bench.build_decoder(spec, dropout=0.0, device=torch.device("cuda"))
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
embed = torch.nn.functional.normalize(torch.randn(B, 512), dim=-1).cuda()
g = torch.Generator().manual_seed(99)
lens = torch.randint(1, 5, (42919,), generator=g)
nouns = torch.randint(1, spec.vocab_size, (42919, spec.token_length), generator=g) * (torch.arange(spec.token_length).unsqueeze(0) < lens.unsqueeze(1))
nouns = torch.unique(nouns, dim=0).cuda()
which = sys.argv[3].split(",") if len(sys.argv) > 3 else ["greedy", "beam4"]
for name, fn in (("greedy", lambda: model.generate(embed, False, True, 1.0, 0.0, None, None, False)),
                 ("beam4", lambda: model.generate_beam(embed, 4, 1.0, 0.0, None, False, 0.0, None, False)),
                 ("beam10g", lambda: model.generate_beam(embed, 10, 1.0, 0.0, None, False, 0.0, nouns, False))):
	if name not in which:
		continue
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

for lanes in ([int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else []):
	embeds = [torch.nn.functional.normalize(torch.randn(B, 512), dim=-1).cuda() for _ in range(lanes)]
	for name, fn in (("greedy", lambda: model.generate_many(embeds, False, True, 1.0, 0.0, None, None, False)),
	                 ("beam4", lambda: model.generate_beam_many(embeds, 4, 1.0, 0.0, None, False, 0.0, None, False)),
	                 ("beam10g", lambda: model.generate_beam_many(embeds, 10, 1.0, 0.0, None, False, 0.0, nouns, False))):
		if name not in which:
			continue
		with torch.no_grad():
			for _ in range(3):
				fn()
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(reps):
				fn()
			t1 = time.perf_counter()
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / reps
		print(f"{name} x {lanes} lanes: {dt * 1e3:.2f} ms per call ({(t1 - t0) / reps * 1e3:.2f} ms of it until the host had enqueued everything), {lanes * B / dt:.0f} labels/s", flush=True)
