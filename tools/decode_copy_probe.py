#!/usr/bin/env python3
"""What the per-step "still active" read-back costs a decode call (round 6): greedy / beam-4 at B rows, 11 forced steps, with
  flag   the session's `advance` as it is now: the last launch of every step's graph writes the answer into mapped host memory (ops.step_done), the host polls;
  copy   what it was: a 4-byte device -> pinned copy + an event behind every step, between the steps' graphs;
  event  the event without the copy;
  none   neither, and no flag launch either is NOT separable here (it is inside the captured graphs) -- so "none" = the flag launch without any host look (no early exit
         possible: the bound of what a free read-back would give);
  k steps per graph: fewer graph launches per call (no host look).
Interleaved rounds inside one process.  python tools/decode_copy_probe.py [B] [rounds]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import embedding_decoder as ED  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
spec = bench.WorkloadSpec(embed_dim=512, vocab_size=6912, token_length=12)
torch.manual_seed(1)
model = bench.build_decoder(spec, dropout=0.0, device=torch.device("cuda"))
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
embed = torch.nn.functional.normalize(torch.randn(B, 512), dim=-1).cuda()

original = ED._DecodeSession.advance


def advance_without_readback(self, C):
	if self.graphs is not None:
		self.graphs[C - 1].replay()
		self.cur = self.cur_after[C - 1]
	else:
		if C == 1:
			self.reset()
		self.cur = self.step(C, self.cur)
	self.final_cur = self.cur
	return C < self.G


def _old_state(self):
	if not hasattr(self, "_probe_events"):
		self._probe_events = [torch.cuda.Event() for _ in range(self.G)]
		self._probe_host = torch.zeros(self.G, dtype=torch.int32).pin_memory()
	return self._probe_events, self._probe_host


def advance_with_copy_and_event(self, C):
	stream = torch.cuda.current_stream()
	events, host = _old_state(self)
	last = self.G if not self.beam else self.G - 1
	if self.graphs is not None:
		self.graphs[C - 1].replay()
		self.cur = self.cur_after[C - 1]
	else:
		if C == 1:
			self.reset()
		self.cur = self.step(C, self.cur)
	self.final_cur = self.cur
	if C <= last:
		host[C - 1:C].copy_(self.active[C - 1:C], non_blocking=True)
		events[C - 1].record(stream)
	if 2 <= C and C - 1 <= last:
		events[C - 2].synchronize()
		if int(host[C - 2]) == 0:
			return False
	return C < self.G


def advance_with_the_event_only(self, C):
	stream = torch.cuda.current_stream()
	self.done_events = _old_state(self)[0]
	last = self.G if not self.beam else self.G - 1
	if self.graphs is not None:
		self.graphs[C - 1].replay()
		self.cur = self.cur_after[C - 1]
	else:
		if C == 1:
			self.reset()
		self.cur = self.step(C, self.cur)
	self.final_cur = self.cur
	if C <= last:
		self.done_events[C - 1].record(stream)
	if 2 <= C and C - 1 <= last:
		self.done_events[C - 2].synchronize()
	return C < self.G


def advance_in_chunks(k):
	"""k consecutive steps per hipGraph, neither copy nor event (what fewer graph launches per call are worth)."""
	def advance(self, C):
		if self.graphs is None:
			return advance_without_readback(self, C)
		key = "_chunks%d" % k
		if not hasattr(self, key):
			from novic_amd import ops
			chunks, side, cur = [], ops.capture_stream(torch.cuda.current_device()), 0
			side.wait_stream(torch.cuda.current_stream())
			with torch.inference_mode(False), torch.cuda.stream(side):
				for c0 in range(1, self.G + 1, k):
					g = torch.cuda.CUDAGraph()
					with ops.graph_capture(g, side):
						for c in range(c0, min(c0 + k, self.G + 1)):
							if c == 1:
								self.reset()
							cur = self.step(c, cur)
					chunks.append((g, cur))
			torch.cuda.current_stream().wait_stream(side)
			setattr(self, key, chunks)
		if (C - 1) % k == 0:
			g, cur = getattr(self, key)[(C - 1) // k]
			g.replay()
			self.cur = cur
		self.final_cur = self.cur
		return C < self.G
	return advance


def timed(fn, reps=20):
	with torch.no_grad():
		for _ in range(3):
			fn()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(reps):
			fn()
		torch.cuda.synchronize()
	return (time.perf_counter() - t0) / reps


fns = {"greedy": lambda: model.generate(embed, False, True, 1.0, 0.0, None, None, False), "beam4": lambda: model.generate_beam(embed, 4, 1.0, 0.0, None, False, 0.0, None, False)}
variants = (("flag", original), ("copy", advance_with_copy_and_event), ("event", advance_with_the_event_only), ("none", advance_without_readback), ("2 steps per graph", advance_in_chunks(2)),
            ("3 steps per graph", advance_in_chunks(3)), ("all steps in one graph", advance_in_chunks(64)))
res = {(n, v): [] for n in fns for v, _ in variants}
for r in range(rounds):
	for v, adv in variants:
		ED._DecodeSession.advance = adv
		for n, fn in fns.items():
			res[(n, v)].append(timed(fn))
ED._DecodeSession.advance = original
for n in fns:
	f, a, e, b = (sorted(res[(n, v)])[rounds // 2] for v in ("flag", "copy", "event", "none"))
	print(f"{n} at {B} rows: {f * 1e3:.3f} ms per call as it is ({B / f:.0f} labels/s); {a * 1e3:.3f} ms with a copy + event per step on top ({B / a:.0f}), {e * 1e3:.3f} ms with the "
	      f"event only ({B / e:.0f}), {b * 1e3:.3f} ms without any host look ({B / b:.0f} labels/s)", flush=True)
	for v, _ in variants[4:]:
		t = sorted(res[(n, v)])[rounds // 2]
		print(f"    {v} (no read-back): {t * 1e3:.3f} ms per call, {B / t:.0f} labels/s", flush=True)
