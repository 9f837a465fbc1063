#!/usr/bin/env python3
"""Host and device timeline of the coalesced ViT-B/32 + greedy decode pipeline WITHOUT a profiler (round 6): per tower launch and per decode call the host's enqueue
window (perf_counter) and the device's execution window (HIP events on the stream that carries it), a few steady-state groups side by side.
python tools/e2e_timeline.py [batches per launch] [decode rows] [budget] [ahead] [big|small] [lanes] [resident|host_u8|host_f32]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import clip_vit, embedders, ops  # noqa: E402
from novic_amd.infer import split_decode_groups  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
cus = int(sys.argv[3]) if len(sys.argv) > 3 and int(sys.argv[3]) > 0 else None
ahead = int(sys.argv[4]) if len(sys.argv) > 4 else 1
big_tiles = len(sys.argv) > 5 and sys.argv[5] == "big"
lanes = int(sys.argv[6]) if len(sys.argv) > 6 else 1  # decode calls of this many consecutive groups AT THE SAME TIME (generate_many: one lane stream each)  # the decode steps on the 128 x 128 GEMM kernel (decode_fused = False) instead of the small-tile launches
dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)
torch.manual_seed(0)
model = bench.build_decoder(spec, dropout=0.0, device=dev)
with torch.no_grad():
	model.logits_linear.weight[0].zero_()
model.eval()
if big_tiles:
	model.decode_fused = False
vit = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).to(dev)
vit.half_stream = True
B = 256
g = torch.Generator().manual_seed(B)
source = sys.argv[7] if len(sys.argv) > 7 else "resident"  # resident | host_u8 | host_f32 (pinned host batches through the stager)
if source == "host_u8":
	res = [torch.randint(0, 256, (B, 3, 224, 224), generator=g, dtype=torch.uint8).pin_memory() for _ in range(12)]
elif source == "host_f32":
	res = [torch.randn(B, 3, 224, 224, generator=g).pin_memory() for _ in range(12)]
else:
	res = [torch.randn(B, 3, 224, 224, generator=g).to(dev) for _ in range(12)]
if os.environ.get("E2E_DISTINCT"):  # fewer distinct batches, repeated (bench.py cycles four)
	k = int(os.environ["E2E_DISTINCT"])
	res = (res[:k] * 12)[:12]
log = []
host_time = {"wait": 0.0, "advance": 0.0, "n": 0}
if os.environ.get("E2E_HOST_TIME"):  # where the host thread's time goes inside the decode calls: waiting for a step's word against everything else of advance() (graph launches)
	from novic_amd import embedding_decoder as ED
	_w, _a = ED._DecodeSession._wait_done, ED._DecodeSession.advance

	def wait_done(self, i):
		t = time.perf_counter()
		v = _w(self, i)
		host_time["wait"] += time.perf_counter() - t
		return v

	def advance(self, C):
		t = time.perf_counter()
		v = _a(self, C)
		host_time["advance"] += time.perf_counter() - t
		host_time["n"] += 1
		return v
	ED._DecodeSession._wait_done, ED._DecodeSession.advance = wait_done, advance
t_origin = [0.0]
ev_origin = torch.cuda.Event(enable_timing=True)


def tower(arg):
	s = torch.cuda.current_stream()
	e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	h0 = time.perf_counter()
	e0.record(s)
	out = vit(arg)
	e1.record(s)
	log.append(("tower", h0, time.perf_counter(), e0, e1))
	return out


def decode(parts, record):
	if record:
		s = torch.cuda.current_stream()
		e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
		h0 = time.perf_counter()
		e0.record(s)
	if len(parts) == 1:
		model.generate(parts[0], False, True, 1.0, 0.0, None, None, False)
	else:
		model.generate_many(parts, False, True, 1.0, 0.0, None, None, False)
	if record:
		e1.record(s)
		log.append((f"decode x{len(parts)}", h0, time.perf_counter(), e0, e1))


def run(batches, record):
	held = []
	for e, sizes in embedders.pipeline_image_batches(tower if record else vit, batches, dev, cus, ahead=ahead, coalesce=n, grouped=True):
		for (a, b), _ in split_decode_groups(sizes, rows):
			held.append(e[a:b])
			if len(held) == lanes:
				decode(held, record)
				held = []
	if held:
		decode(held, record)


with torch.no_grad():
	for _ in range(3):
		run(res, False)
	torch.cuda.synchronize()
	host_time.update(wait=0.0, advance=0.0, n=0)
	t0 = time.perf_counter()
	LONG = int(os.environ.get("E2E_LONG", "4"))  # x 12 batches of 256 in the timed run (the pipeline's fill and drain are a fixed cost: short runs flatter or punish it)
	run(res * LONG, False)
	torch.cuda.synchronize()
	if host_time["n"]:
		print(f"host time in the decode calls of the timed run: {host_time['n']} advance() calls, {1e3 * host_time['advance']:.1f} ms in all, of which {1e3 * host_time['wait']:.1f} ms waiting for a "
		      f"step's word; the run took {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
	print(f"unrecorded: {B * len(res) * LONG / (time.perf_counter() - t0) / 1e3:.1f} k labels/s (coalesce {n}, decode rows {rows}, budget {cus}, ahead {ahead}, {'128 x 128 tiles' if big_tiles else 'small tiles'}, {lanes} decode lane(s), {source})", flush=True)
	torch.cuda.synchronize()
	ev_origin.record(torch.cuda.current_stream())
	t_origin[0] = time.perf_counter()
	run(res * 3, True)
	torch.cuda.synchronize()
	print(f"recorded:   {B * len(res) * 3 / (time.perf_counter() - t_origin[0]) / 1e3:.1f} k labels/s", flush=True)
if os.environ.get("E2E_QUIET"):
	sys.exit(0)
print("kind     host enqueue window (ms)      device execution window (ms)")
for kind, h0, h1, e0, e1 in log[len(log) // 3: len(log) // 3 + 14]:
	print(f"{kind:9s}  {1e3 * (h0 - t_origin[0]):8.2f} .. {1e3 * (h1 - t_origin[0]):8.2f}      {ev_origin.elapsed_time(e0):8.2f} .. {ev_origin.elapsed_time(e1):8.2f}   ({e0.elapsed_time(e1):6.2f} ms)")
