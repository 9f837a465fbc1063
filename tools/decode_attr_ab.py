#!/usr/bin/env python3
"""A/B of a routing attribute of the decode step -- two models that differ in ONE class attribute of PrefixedIterDecoder -- greedy / beam-4 at the bench's sizes, interleaved
rounds in one process; checks that the outputs are bit-identical.  python tools/decode_attr_ab.py [ATTR V0 V1]   (default: decode_ffn_fused 0 1 -- the feed-forward half of a
layer as two launches against one, novic_decode_ffn; e.g. decode_ln_rows 512 1024)"""
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

spec = bench.WorkloadSpec(embed_dim=512, vocab_size=6912, token_length=12)
torch.manual_seed(1)
models = {}
ATTR = sys.argv[1] if len(sys.argv) > 1 else "decode_ffn_fused"
VALS = [int(v) for v in sys.argv[2:4]] if len(sys.argv) > 3 else [0, 1]
for fused, val in zip((False, True), VALS):
	m = bench.build_decoder(spec, dropout=0.0, device=torch.device("cuda"))
	with torch.no_grad():
		m.logits_linear.weight[0].zero_()
	m.eval()
	assert hasattr(m, ATTR), ATTR
	setattr(m, ATTR, type(getattr(m, ATTR))(val))
	models[fused] = m
models[True].load_state_dict(models[False].state_dict())
for name, B, fn in (("greedy", 256, lambda m, e: m.generate(e, True, True, 1.0, 0.0, None, None, False)),
                    ("greedy", 1024, lambda m, e: m.generate(e, True, True, 1.0, 0.0, None, None, False)),
                    ("beam-4", 256, lambda m, e: m.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)),
                    ("beam-4", 384, lambda m, e: m.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)),
                    ("beam-4", 512, lambda m, e: m.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)),
                    ("beam-4", 768, lambda m, e: m.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)),
                    ("beam-4", 1024, lambda m, e: m.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)),
                    ("greedy", 2048, lambda m, e: m.generate(e, True, True, 1.0, 0.0, None, None, False))):
	e = torch.nn.functional.normalize(torch.randn(B, 512), dim=-1).cuda()
	res, outs = {False: [], True: []}, {}
	with torch.no_grad():
		for fused, m in models.items():
			for _ in range(3):
				outs[fused] = fn(m, e)
		torch.cuda.synchronize()
		for rnd in range(5):
			for fused, m in models.items():
				t0 = time.perf_counter()
				for _ in range(8):
					fn(m, e)
				torch.cuda.synchronize()
				res[fused].append((time.perf_counter() - t0) / 8)
	same = all((a is None and b is None) or torch.equal(a, b) for a, b in zip(outs[False], outs[True]))
	print(f"{name} B {B}: {ATTR} = {VALS[0]}: {B / statistics.median(res[False]) / 1e3:.1f} k labels/s, = {VALS[1]}: {B / statistics.median(res[True]) / 1e3:.1f} k; outputs bit-identical: {same}", flush=True)
