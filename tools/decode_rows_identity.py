#!/usr/bin/env python3
"""Does a sample's decode result depend on how many rows share its decode call?  Greedy / beam-4 at the bench's size: the first 256 samples decoded alone and as the head of
calls of 512 / 768 / 1 024 / 2 048 rows; prints which outputs are bit-identical.  python tools/decode_rows_identity.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
g = torch.Generator().manual_seed(5)
e = torch.nn.functional.normalize(torch.randn(2048, spec.embed_dim, generator=g), dim=-1).to(dev)
with torch.no_grad():
	for name, fn, keep in (("greedy", lambda x: model.generate(x, False, True, 1.0, 0.0, None, None, False), (0, 1, 5)),
	                       ("beam-4", lambda x: model.generate_beam(x, 4, 1.0, 0.0, None, False, 0.0, None, False), (0, 1, 2))):
		ref = [None if t is None else t.clone() for t in fn(e[:256].contiguous())]
		for n in (512, 768, 1024, 2048):
			for rep in range(2):  # eager, then graph replay
				out = fn(e[:n].contiguous())
			same = [bool(torch.equal(out[i][:256], ref[i])) for i in keep]
			ids_diff = int((out[0][:256] != ref[0]).any(dim=-1).sum()) if out[0].ndim == 2 else int((out[0][:256] != ref[0]).flatten(1).any(dim=1).sum())
			sc = keep[2]
			print(f"{name}: 256 rows alone vs head of {n}: ids / padding / score bit-identical: {same}; samples whose ids differ: {ids_diff}; max |dscore| {float((out[sc][:256] - ref[sc]).abs().max()):.3g}", flush=True)

# the released default: guided beam-10 over a noun vocabulary (early exit when every beam of every sample has spelt a noun: a larger call may run more steps, so the
# shorter result must equal the longer one's leading columns, the rest being padding)
g2 = torch.Generator().manual_seed(99)
lens = torch.randint(1, 5, (42919,), generator=g2)
nouns = torch.randint(1, spec.vocab_size, (42919, spec.token_length), generator=g2) * (torch.arange(spec.token_length).unsqueeze(0) < lens.unsqueeze(1))
nouns = torch.unique(nouns, dim=0).to(dev)
with torch.no_grad():
	fn = lambda x: model.generate_beam(x, 10, 1.0, 0.0, None, False, 0.0, nouns, False)
	ref = [t.clone() for t in fn(e[:256].contiguous())]
	for n in (512, 1024):
		for rep in range(2):
			out = fn(e[:n].contiguous())
		T = min(ref[0].shape[-1], out[0].shape[-1])
		same_ids = bool(torch.equal(out[0][:256, :, :T], ref[0][..., :T])) and bool((out[0][:256, :, T:] == 0).all()) and bool((ref[0][..., T:] == 0).all())
		same_pad = bool(torch.equal(out[1][:256, :, :T], ref[1][..., :T]))
		print(f"guided beam-10: 256 samples alone ({ref[0].shape[-1]} steps) vs head of {n} ({out[0].shape[-1]} steps): ids {same_ids}, padding {same_pad}, scores bit-identical "
		      f"{bool(torch.equal(out[2][:256], ref[2]))} (max |d| {float((out[2][:256] - ref[2]).abs().max()):.3g})", flush=True)

# where the greedy regimes part: per-step logits of the first 256 samples, alone and as the head of 1 024 rows
with torch.no_grad():
	a = model.generate(e[:256].contiguous(), True, True, 1.0, 0.0, None, None, False)[2]
	b = model.generate(e[:1024].contiguous(), True, True, 1.0, 0.0, None, None, False)[2][:256]
	for t in range(a.shape[1]):
		d = (a[:, t].float() - b[:, t].float()).abs()
		print(f"step {t + 1}: rows whose logits differ {int((d.max(dim=1).values > 0).sum())} of 256, max |d| {float(d.max()):.4g}", flush=True)
