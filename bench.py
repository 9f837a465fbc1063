#!/usr/bin/env python3
"""Benchmark of the NOVIC decoder hot path on MI355X (BASELINE.json: decoder train samples/s + infer labels/s).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is ONE OPTIMIZER STEP of the 6-layer / d=512 decoder on synthetic cached-text-embedding data: `accum` = 16 micro-batches of
512 samples per GPU (reference recipe README.md:322 / config/train.yaml: batch 512, AdamW(0.9, 0.95), wd 0.1, clip 1.0, dropout 0.1,
noise GaussElemUniformAngle 3.25 / 45-75 deg / 0.15), i.e. noise + forward + backward of 8192 samples + gradient all-reduce + clip +
AdamW, bf16 MFMA GEMMs with fp32 accumulation.  value = samples/s over all ranks.  Inputs are resident in HBM before timing.
Rank 0 prints one JSON line; extra keys carry the decode throughput (labels/s), the roofline of the dominant kernel measured live
with HIP events, and the CPU baseline (the oracle port on the host cores).
"""
from __future__ import annotations

import argparse
import dataclasses
import json
import math
import os
import sys
import time

if int(os.environ.get("WORLD_SIZE", "1") or "1") <= 1:  # (single process only: untested beside RCCL -- novic_amd/__init__.py; the lane legs are skipped at N > 1)
	os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # before the HIP runtime starts: lanes on streams of their own need hardware queues of their own

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_BYTES_PER_S = 8.0e12    # HBM3E, same guide
HBM_PEAK_GBS = 8000.0

# workload (SURVEY.md 8d / BASELINE.md 3)
F_DIM, VOCAB, CMAX, MICRO_B, ACCUM = 512, 6912, 12, 512, 16
MAX_CONTENT = 6  # content tokens per label ~ U{1..6} (+ END) -> C = 7, S = 10


_T0 = time.perf_counter()


def note(msg):
	"""Progress line on stderr (the JSON result is the only thing on stdout)."""
	if int(os.environ.get("RANK", "0")) == 0:
		print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


def host_threads() -> int:
	try:
		n = len(os.sched_getaffinity(0))
	except AttributeError:
		n = os.cpu_count() or 1
	return max(1, min(n, 16))  # the GPU box gives one GPU's job a 16-core share


class ClockSampler:
	"""Shader clock of this rank's GPU while the timed regions run: a thread reads the amdgpu hwmon file of the device (freq1_input, Hz) every 20 ms -- a sysfs read, no GPU
	call and no other process -- and stop() returns {'mhz_min', 'mhz_median', 'mhz_max', 'samples', 'source'} (or {'source': None} where the file is not readable)."""

	def __init__(self, device):
		import glob
		import threading
		self.path = None
		try:
			props = torch.cuda.get_device_properties(device)
			bdf = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
			hits = glob.glob(f"/sys/bus/pci/devices/{bdf}/hwmon/hwmon*/freq1_input")
			if not hits and torch.cuda.device_count() == 1:
				hits = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
			self.path = hits[0] if hits else None
		except Exception:
			self.path = None
		self.samples, self._stop, self._thread = [], threading.Event(), None

	def _read(self):
		try:
			with open(self.path) as f:
				return int(f.read().strip()) / 1e6
		except Exception:
			return None

	def start(self):
		import threading
		if self.path is None:
			return

		def run():
			while not self._stop.is_set():
				v = self._read()
				if v is not None:
					self.samples.append(v)
				self._stop.wait(0.02)
		self._thread = threading.Thread(target=run, name="bench-clock", daemon=True)
		self._thread.start()

	def stop(self) -> dict:
		if self._thread is None:
			return {"source": None}
		self._stop.set()
		self._thread.join()
		if not self.samples:
			return {"source": None}
		xs = sorted(self.samples)
		return {"mhz_min": round(xs[0]), "mhz_median": round(xs[len(xs) // 2]), "mhz_max": round(xs[-1]), "samples": len(xs), "source": self.path}


def parse():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=20)
	ap.add_argument("--warmup", type=int, default=3)
	ap.add_argument("--accum", type=int, default=ACCUM)
	ap.add_argument("--repeats", type=int, default=7, help="the timed region (exactly --steps steps between barrier + synchronize on both sides) is run this many times back to "
	                "back after the warm-up; `ms_per_step` / `value` are the MEDIAN region's, min / max and every region are reported beside it (a 0.13 s region on a box "
	                "with a +-3 %% spread cannot register a 1 %% kernel change on its own)")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-decode", action="store_true")
	ap.add_argument("--no-dense", action="store_true", help="skip the every-position-computed variant of the step (profile collection: the trace then ends with the timed steps)")
	ap.add_argument("--decode-batch", type=int, default=256)
	ap.add_argument("--persistent-cus", type=str, default=None, help="N > 1 only: workgroups the persistent GEMM grids of the backward pass may have while the early per-layer "
	                "all-reduces are in flight (train.DataParallel(persistent_cus)); default: all 256.  For A/B runs on a multi-GPU node: the grids otherwise own every CU beside RCCL's "
	                "kernels.  'A,B[,C]': the line's `value` is measured with A, and one more timed region + one instrumented pass runs per further budget (`dp_budget_ab`); 0 = no reservation")
	ap.add_argument("--fingerprint", action="store_true", help="print the source fingerprint the traffic figures are tied to and exit (no GPU call)")
	args = ap.parse_args()
	budgets = [int(x) for x in str(args.persistent_cus).split(",") if x.strip() != ""] if args.persistent_cus not in (None, "") else []
	args.persistent_cus_list = [b if b > 0 else None for b in budgets]
	args.persistent_cus = args.persistent_cus_list[0] if budgets else None
	return args


def synth_micro_batch(spec, B, seed, device):
	g = torch.Generator().manual_seed(seed)
	embed = torch.nn.functional.normalize(torch.randn(B, spec.embed_dim, generator=g), dim=-1)
	lens = torch.randint(1, MAX_CONTENT + 1, (B,), generator=g)
	C = MAX_CONTENT + 1
	col = torch.arange(C).unsqueeze(0)
	target = torch.randint(1, spec.vocab_size, (B, C), generator=g) * (col < lens.unsqueeze(1))
	pad = col > lens.unsqueeze(1)
	return embed.to(device), target.to(device), pad.to(device), None


def flops_per_sample_train(spec, S, T, S2=None):
	"""S = sequence positions per sample that are computed, S2 = the mean of their square (attention), T = output positions per sample whose logits are
	computed -- floats when the padded ones are skipped (packed rows / compacted loss block); dense: S, S*S, C."""
	E, K, L, P, F, V = spec.hidden_dim, spec.feedfwd_dim, spec.num_layers, spec.mlp_seq_len, spec.embed_dim, spec.vocab_size
	fwd = 2 * F * P * E + L * (S * (8 * E * E + 4 * E * K) + 4 * (S * S if S2 is None else S2) * E) + 2 * E * V * T
	return int(3 * fwd)


@dataclasses.dataclass(frozen=True)
class WorkloadSpec:
	"""The decoder of config/train.yaml:254-270 (6 layers, d=512, feed-forward 128, 8 heads, 4 prefix tokens)."""
	embed_dim: int
	vocab_size: int
	token_length: int
	hidden_dim: int = 512
	feedfwd_dim: int = 128
	num_layers: int = 6
	num_heads: int = 8
	mlp_seq_len: int = 4


class _CachedTextEmbedder:
	"""The embedder fields the decoder constructor reads (reference embedding_decoder.py:77-86); the bench feeds cached embeddings directly."""

	def __init__(self, spec):
		from novic_amd import embedders
		self.embed_dtype, self.embed_dim, self.target_vocab = torch.float32, spec.embed_dim, ()
		self.target_config = embedders.TargetConfig(vocab_size=spec.vocab_size, token_dtype=torch.int64, mask_dtype=torch.bool, start_token_id=None, end_token_id=0, pad_token_id=0,
		                                            compact_ids=True, compact_map=None, compact_unmap=None, fixed_token_length=False, token_length=spec.token_length, use_masks=True)


def build_decoder(spec, dropout, device, multi_length=1):
	"""PrefixedIterDecoder with the constructor kwargs of reference infer.py:721-758 / config/train.yaml defaults, random init.
	multi_length > 1: the multiset data configuration (several weighted targets per embedding, embedding_dataset.py:20)."""
	from novic_amd import embedding_dataset, embedding_decoder
	multi = multi_length > 1
	dc = embedding_dataset.DataConfig.create(dict(use_weights=multi, unit_weights=True, multi_target=multi, multi_first=False, full_targets=True, fixed_multi_length=True,
	                                               multi_length=multi_length))
	model = embedding_decoder.PrefixedIterDecoder(
		embedder=_CachedTextEmbedder(spec), data_config=dc, vocab_quant=False, num_end_loss=1, label_smoothing=0.0, hidden_dim=spec.hidden_dim,
		feedfwd_scale=f"{spec.feedfwd_dim}/{spec.hidden_dim}", mlp_hidden_layer="none", mlp_hidden_bias=False, mlp_hidden_norm=False, mlp_hidden_activation="gelu", input_dropout=dropout,
		num_layers=spec.num_layers, num_heads=spec.num_heads, layer_dropout=dropout, layer_activation="gelu", layer_norm_first=True, layer_bias=False, logits_bias=False,
		init_bias_zero=True, init_mlp_mode="balanced", init_mlp_unit_norm=False, init_tfrm_mode="balanced", init_tfrm_unit_norm=False, init_tfrm_unit_postnorm=True,
		init_tfrm_proj_layers=True, init_zero_norm=False, init_rezero_mode="none", mlp_seq_len=spec.mlp_seq_len, weight_tying=True, strictly_causal=False, enable_nested=False)
	return model.to(device)


def count_gpu_nodes() -> int:
	"""GPUs of this node as the kernel driver lists them (KFD topology: a node with SIMDs is a GPU, a CPU node has simd_count 0) -- read from sysfs, so the process
	that asks makes no HIP call (torch.cuda.device_count() falls through to hipGetDeviceCount without amdsmi, which initialises the runtime)."""
	import glob
	if not os.path.isdir("/sys/class/kfd"):
		return 0  # no amdgpu compute driver on this machine at all
	n, unreadable = 0, 0
	for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
		try:
			with open(path) as f:
				props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
		except OSError:
			unreadable += 1
			continue
		if int(props.get("simd_count", "0")) > 0:
			n += 1
	return -1 if (n == 0 and unreadable) else n  # -1: the topology is there but not readable by this user -- unknown, the ranks will find out


def launch_ranks(args) -> int:
	"""`python bench.py --gpus N` (N > 1) started directly, the way the driver starts `--gpus 1`: this process becomes the PARENT of an N-rank run.  It makes
	no GPU call at all (the device count comes from sysfs; -1 = topology unreadable = unknown, and then the ranks find out themselves), starts
	`python -m torch.distributed.run` on 127.0.0.1 as a CHILD PROCESS (never an exec) with this file and the same flags, relays rank 0's JSON line and returns the
	child's exit code."""
	import socket
	import subprocess
	rehearse = os.environ.get("NOVIC_BENCH_REHEARSE", "0") == "1"
	have = count_gpu_nodes()
	if 0 <= have < args.gpus and not rehearse:
		print(f"bench.py: --gpus {args.gpus} but this node exposes {have} GPU(s) (NOVIC_BENCH_REHEARSE=1 walks the N-rank control flow on one GPU over gloo)", file=sys.stderr)
		return 2
	with socket.socket() as sock:
		sock.bind(("127.0.0.1", 0))
		port = sock.getsockname()[1]
	cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
	       os.path.abspath(__file__)] + sys.argv[1:]
	env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
	note(f"starting {args.gpus} ranks: {' '.join(cmd[1:])}")
	proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)  # stderr (progress notes of rank 0) passes straight through
	lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
	if lines:
		print(lines[-1], flush=True)
	else:
		sys.stderr.write(proc.stdout)
		return proc.returncode or 1
	return proc.returncode


def main():
	args = parse()
	if args.fingerprint:
		print(source_fingerprint())
		return
	if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
		raise SystemExit(launch_ranks(args))
	rank = int(os.environ.get("RANK", "0"))
	local_rank = int(os.environ.get("LOCAL_RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if not torch.cuda.is_available():
		raise SystemExit("bench.py needs MI355X GPUs (torch.cuda.is_available() is False); there is no CPU fallback for the product path")
	# NOVIC_BENCH_REHEARSE=1: every rank on cuda:0 with the gloo backend -- walks the N > 1 control flow on a one-GPU box (numbers meaningless)
	rehearse = os.environ.get("NOVIC_BENCH_REHEARSE", "0") == "1"
	if rehearse:
		local_rank = 0
	torch.cuda.set_device(local_rank)
	device = torch.device("cuda", local_rank)
	import torch.distributed as dist
	if world > 1:
		os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
		if rehearse:
			dist.init_process_group(backend="gloo")
		else:
			dist.init_process_group(backend="nccl", device_id=device)
	if world != args.gpus:
		raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree, refusing to report a number under the wrong n_gpus")
	ranks_seen = dist.get_world_size() if world > 1 else 1
	if ranks_seen != world:
		raise SystemExit(f"bench.py: the process group has {ranks_seen} ranks, WORLD_SIZE={world}")

	backend_name = "none (single process)"
	if world > 1:
		try:
			backend_name = "gloo (rehearsal on one GPU)" if rehearse else "nccl = RCCL " + ".".join(str(v) for v in torch.cuda.nccl.version())
		except Exception:  # the version query is informational only
			backend_name = "nccl = RCCL"

	from novic_amd import train as T, embedding_noise, ops

	torch.set_num_threads(host_threads())
	spec = WorkloadSpec(embed_dim=F_DIM, vocab_size=VOCAB, token_length=CMAX)
	torch.manual_seed(0)
	note(f"building model + synthetic pool (world {world}, host threads {torch.get_num_threads()})")
	model = build_decoder(spec, dropout=0.1, device=device)
	dp = T.DataParallel(persistent_cus=args.persistent_cus)
	dp.broadcast_parameters(model.flat_parameters())
	model.train()
	opt = T.FusedAdamW(model, lr=1.5e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	noise = embedding_noise.EmbeddingNoise.create("GaussElemUniformAngle", F_DIM, 3.25, 45.0, 75.0, 0.0, 0.15)
	dp.decorrelate(model, noise)  # per-rank dropout / noise streams
	accum = args.accum
	# a pool of 2 distinct optimizer steps' worth of micro-batches per rank, resident in HBM (noise works in place -> cloned per step)
	pool = [[synth_micro_batch(spec, MICRO_B, 1234 + rank * 1000 + s * accum + j, device) for j in range(accum)] for s in range(2)]
	pool_embed = [torch.stack([mb[0] for mb in step]) for step in pool]  # accum x MICRO_B x F per pooled step: one copy per step instead of `accum`
	# output positions that are not padding, per pooled step (the loss block computes only those; counted here, before anything is timed, for the
	# roofline's FLOP accounting): averaged over the steps the timed region will run
	valid_rows = [float(sum(int((~mb[2]).sum()) for mb in step)) for step in pool]
	over_steps = lambda per_pool: sum(per_pool[i % len(pool)] for i in range(args.steps)) / max(1, args.steps)
	rows_computed = over_steps(valid_rows) if getattr(model, "compact_outputs", False) else float(MICRO_B * accum * (MAX_CONTENT + 1))
	# sequence positions that exist in the packed-row layout: the prefix + the label tokens that are inputs -- input position P + c is padded iff target
	# position c + 1 is (embedding_decoder.py:696-712 with num_end_loss = 1: END is only ever predicted), so a sample keeps P - 1 + #unpadded targets
	P_, S_ = spec.mlp_seq_len, spec.mlp_seq_len + MAX_CONTENT
	packed = getattr(model, "pack_rows", False) and getattr(model, "compact_outputs", False)
	seq_len_host = [torch.cat([P_ - 1 + (~mb[2]).sum(dim=1) for mb in step]).double().cpu() for step in pool]
	pos_per_sample = over_steps([float(x.mean()) for x in seq_len_host]) if packed else float(S_)
	pos_sq_per_sample = over_steps([float((x * x).mean()) for x in seq_len_host]) if packed else float(S_ * S_)

	# The micro-batches of a pooled step are handed over the way the product's loader hands them over (embedding_cache.DeviceLoader(group = accum), what action_train runs):
	# as the slices of ONE set of step buffers (GroupSlice), so that train_step takes the buffers whole instead of concatenating sixteen micro-batches per step -- targets
	# and masks stacked once here, the embeddings cloned per step (the noise works in place).
	from novic_amd.embedding_cache import GroupSlice
	pool_target = [torch.cat([mb[1] for mb in step]) for step in pool]
	pool_mask = [torch.cat([mb[2] for mb in step]) for step in pool]

	def one_step(i):
		k = i % len(pool)
		fresh = pool_embed[k].clone()
		full = (fresh.view(-1, fresh.shape[-1]), pool_target[k], pool_mask[k], None)
		mbs = [GroupSlice((fresh[j], t, m, w), full, j, accum) for j, (_, t, m, w) in enumerate(pool[k])]
		return T.train_step(model, opt, mbs, embed_noise=noise, dp=dp)

	note("warmup")
	for i in range(args.warmup):
		one_step(i)
		torch.cuda.synchronize()
		note(f"  warmup step {i} done")
	torch.cuda.synchronize()
	note("timed region")
	# R regions of exactly K steps each, every one bracketed by barrier + synchronize on both sides and reduced to the MAX over ranks; the line's `ms_per_step` / `value`
	# are the median region's.  A sampler thread reads the shader clock out of sysfs every 20 ms meanwhile (a file read, no GPU call, no extra process).
	clock = ClockSampler(device)
	clock.start()
	regions = []
	for r in range(max(1, args.repeats)):
		if world > 1:
			dist.barrier()
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for i in range(args.steps):
			stats, gnorm = one_step(i)
		torch.cuda.synchronize()
		if world > 1:
			dist.barrier()
		torch.cuda.synchronize()
		el = time.perf_counter() - t0
		if world > 1:
			tmax = torch.tensor([el], dtype=torch.float64, device=device)
			dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
			el = float(tmax)
		regions.append(el)
	clock_info = clock.stop()
	elapsed = sorted(regions)[len(regions) // 2]  # median (the upper one of an even count)
	# The roofline's launch durations: the SAME K steps once more (same pooled batches, same order), now with a HIP event pair around every launch of the priced kernel
	# classes, on the stream they are launched on.  Kept out of the timed region above: ~22 event pairs per step sit between kernels that otherwise run back to back
	# and cost the step 2-4 % (measured), which `value` must not carry; `events_ms_per_step` reports what the instrumented steps took.
	model.logits_gemm_timer = []  # the largest forward GEMM's launch
	model.wgrad_timer = []        # every 256-wide weight-gradient launch pair
	model.gemm_timer = []         # every large K-contiguous GEMM of the 256 x 256 tile kernel (QKV, logits, their input gradients): the largest class of the step
	torch.cuda.synchronize()
	te = time.perf_counter()
	for i in range(args.steps):
		one_step(i)
	torch.cuda.synchronize()
	events_ms = 1000 * (time.perf_counter() - te) / max(1, args.steps)
	logits_events, model.logits_gemm_timer = model.logits_gemm_timer, None
	wgrad_events, model.wgrad_timer = model.wgrad_timer, None
	gemm_events, model.gemm_timer = model.gemm_timer, None
	if world > 1:
		dist.barrier()
	# Data-parallel exchange, first-contact instrumentation (N > 1; never inside the timed regions above): the same K steps with an event pair around the end of the exchange
	# (stream time between the last backward kernel and the optimizer launch = the part of the all-reduce the early per-layer reductions did not hide) and the bytes that
	# went out early / in the tail -- for the budget of the line and, with `--persistent-cus A,B`, one timed region + one such pass per further budget.
	dp_report = None
	if world > 1:
		def dp_pass(cus, timed):
			dp.persistent_cus = cus
			out = {"persistent_cus": cus}
			if timed:
				dist.barrier(); torch.cuda.synchronize()
				t0 = time.perf_counter()
				for i in range(args.steps):
					one_step(i)
				torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
				tm = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
				dist.all_reduce(tm, op=dist.ReduceOp.MAX)
				out["ms_per_step"] = round(1000 * float(tm) / args.steps, 3)
			dp.reset_instrument(True)
			for i in range(args.steps):
				one_step(i)
			exposed = dp.exposed_ms_per_step()
			out.update(allreduce_exposed_ms_per_step=None if exposed is None else round(exposed, 4), allreduce_bytes_early_per_step=dp.bytes_early // max(1, args.steps),
			           allreduce_bytes_tail_per_step=dp.bytes_tail // max(1, args.steps))
			dp.reset_instrument(False)
			return out
		first = dp_pass(args.persistent_cus, timed=False)
		first["ms_per_step"] = round(1000 * elapsed / args.steps, 3)
		dp_report = [first] + [dp_pass(c, timed=True) for c in args.persistent_cus_list[1:]]
		dp.persistent_cus = args.persistent_cus
		dist.barrier()
	samples = MICRO_B * accum * world * args.steps
	value = samples / elapsed
	# The same optimizer step with EVERY position computed (the reference's dense layout: padded positions are run through the layers and the loss
	# block and then masked): reported next to `value` so that both readings of "one step" are on the record.  Same barrier / max-over-ranks timing.
	dense_value = None
	if (getattr(model, "pack_rows", False) or getattr(model, "compact_outputs", False)) and not args.no_dense:
		cls = type(model)
		saved_flags = (cls.pack_rows, cls.compact_outputs)
		cls.pack_rows = cls.compact_outputs = False
		try:
			n_dense = max(1, min(args.steps, 10))
			for i in range(2):
				one_step(i)
			torch.cuda.synchronize()
			if world > 1:
				dist.barrier()
			torch.cuda.synchronize()
			td = time.perf_counter()
			for i in range(n_dense):
				one_step(i)
			torch.cuda.synchronize()
			if world > 1:
				dist.barrier()
			torch.cuda.synchronize()
			dense_elapsed = time.perf_counter() - td
			if world > 1:
				tmax = torch.tensor([dense_elapsed], dtype=torch.float64, device=device)
				dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
				dense_elapsed = float(tmax)
			dense_value = MICRO_B * accum * world * n_dense / dense_elapsed
			note(f"train, every position computed: {dense_value:.0f} samples/s ({1000 * dense_elapsed / n_dense:.2f} ms/step)")
		finally:
			cls.pack_rows, cls.compact_outputs = saved_flags
	note(f"train: {value:.0f} samples/s ({1000 * elapsed / args.steps:.2f} ms/step)")
	loss = float((stats[1] / stats[0]).mean())
	assert math.isfinite(loss) and math.isfinite(float(gnorm))

	result = None
	# The spread of the timed regions and the clock they ran at, inside `config` (the part of the line the driver's record keeps whole) and once more as the LAST keys of
	# the line: `cycles_per_step` = ms_per_step x the median shader clock is the figure that compares across boxes (they differ by up to 5 % with the clock they hold).
	ms_med = 1000 * elapsed / args.steps
	timing = {"how": f"median of {len(regions)} back-to-back regions of exactly {args.steps} steps, each between barrier + synchronize, max over ranks",
	          "repeats": len(regions), "ms_per_step_min": round(1000 * min(regions) / args.steps, 3), "ms_per_step_max": round(1000 * max(regions) / args.steps, 3),
	          "ms_per_step_regions": [round(1000 * x / args.steps, 3) for x in regions], "shader_clock_mhz": {k: clock_info.get(k) for k in ("mhz_min", "mhz_median", "mhz_max")},
	          "cycles_per_step": None if not clock_info.get("mhz_median") else int(round(ms_med * 1e-3 * clock_info["mhz_median"] * 1e6))}
	if rank == 0:
		S, Tt = spec.mlp_seq_len + MAX_CONTENT, MAX_CONTENT + 1
		fl = flops_per_sample_train(spec, pos_per_sample, rows_computed / (MICRO_B * accum), pos_sq_per_sample)  # the FLOP actually issued: non-padded positions only
		result = {
			"metric": "decoder train samples/s + infer labels/s (ViT-B/32, 6L dec) at 1/2/4/8 GPU",
			"value": round(value, 1), "unit": "samples/s", "n_gpus": world, "n_ranks_seen": ranks_seen, "collective_backend": backend_name, "steps": args.steps, "warmup": args.warmup,
			"ms_per_step": round(1000 * elapsed / args.steps, 3),
			"higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
			"config": {"workload": "6L/d512 embedding_decoder training step on cached ViT-B/32 text embeddings + noise (configs[1])", "micro_batch": MICRO_B, "accum": accum,
			           "timing": timing,
			           "global_batch": MICRO_B * accum * world, "embed_dim": F_DIM, "vocab": VOCAB, "seq_len": S, "label_tokens": Tt, "dropout": 0.1,
			           "noise": "GaussElemUniformAngle(3.25,45-75deg,0.15)", "optimizer": "AdamW(0.9,0.95) wd0.1 clip1.0", "parallelism": f"dp{world}", "dp_persistent_cus": dp.persistent_cus,
			           "padded_positions": f"zero loss and gradient, not computed: {pos_per_sample:.2f} of {S} sequence positions per sample in the layers (packed rows), "
			                               f"logits / cross-entropy for {rows_computed:.0f} of {MICRO_B * accum * Tt} output positions per step"},
			"train_loss_last": round(loss, 4),
			"train_all_positions_samples_per_s": None if dense_value is None else round(dense_value, 1),  # padded positions computed and masked, as the reference does
			"train_mfma_frac_whole_step": round(value / world * fl / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4),
			"train_flop_per_sample": fl,
			"roofline_events": {"where": f"a second pass of the same {args.steps} steps behind the timed region, HIP event pairs around every priced launch",
			                    "events_ms_per_step": round(events_ms, 3)},
		}
		if dp_report is not None:
			result.update(allreduce_exposed_ms_per_step=dp_report[0]["allreduce_exposed_ms_per_step"], allreduce_bytes_early_per_step=dp_report[0]["allreduce_bytes_early_per_step"],
			              allreduce_bytes_tail_per_step=dp_report[0]["allreduce_bytes_tail_per_step"], dp_budget_ab=dp_report,
			              allreduce_note="exposed = stream time between the last backward kernel and the optimizer launch (HIP events, a pass of its own behind the timed regions)"
			                             + ("; gloo REHEARSAL on one GPU: the numbers are meaningless, only the keys are exercised" if rehearse else ""))
		# the whole step against the OTHER roofline: HBM bytes per optimizer step from the committed PMC passes of this command (profiles/r02_hbm_per_step.csv;
		# a property of the kernels and the batch, not of the run) over this run's step time, as a fraction of 8 TB/s
		step_bytes, step_note = _profile_traffic("train_step_hbm_bytes")
		result["train_hbm_profile"] = step_note
		if step_bytes:
			result["train_hbm_GB_per_step_profiled"] = round(step_bytes / 1e9, 2)
			result["train_hbm_frac_whole_step"] = round(step_bytes / (elapsed / args.steps) / HBM_PEAK_BYTES_PER_S, 4)
		packed_rows = pos_per_sample * MICRO_B * accum  # sequence positions the layers run per step (K of the layer weight gradients)
		result["roofline"] = gemm_class_roofline(model, spec, gemm_events, packed_rows, rows_computed, events_ms, args.steps)
		result["roofline_wgrad"] = wgrad_roofline(model, spec, wgrad_events, packed_rows, events_ms, args.steps, rows_computed)
		result["roofline_best_gemm"] = measure_roofline(model, spec, device, ops, logits_events, rows_computed)
		note(f"roofline: {result['roofline']}")
		note(f"roofline_wgrad: {result['roofline_wgrad']}")
		note(f"roofline_best_gemm: {result['roofline_best_gemm']}")
	if not args.no_decode and world == 1:
		model._ws.clear()
		tl = measure_train_loop(device, accum, value)
		note(f"train loop: {tl}")
		result.update(tl)
	if not args.no_decode:
		model._ws.clear()  # the headline step's activations: the legs below bring their own
		ms = measure_multiset(device, rank, world, dist if world > 1 else None, accum, persistent_cus=args.persistent_cus)
		note(f"multiset step: {ms}")
		if rank == 0:
			result.update(ms)
		dec = measure_decode(spec, device, args.decode_batch, world, dist if world > 1 else None)
		note(f"decode: {dec}")
		if rank == 0:
			result.update(dec)
	if rank == 0:
		if world == 1 and not args.no_cpu_baseline:
			note("cpu baseline (oracle port)")
			result["cpu_baseline"] = cpu_baseline(spec)
		# the spread once more as the last keys of the line (whatever keeps only the end of it keeps these), then the line, then a short summary on stderr
		result.update(ms_per_step_min=timing["ms_per_step_min"], ms_per_step_max=timing["ms_per_step_max"], repeats=timing["repeats"], ms_per_step_regions=timing["ms_per_step_regions"],
		              shader_clock=clock_info, cycles_per_step=timing["cycles_per_step"])
		print(json.dumps(result), flush=True)
		keys = ("value", "ms_per_step", "ms_per_step_min", "ms_per_step_max", "cycles_per_step", "train_hbm_GB_per_step_profiled", "infer_greedy_labels_per_s", "infer_beam4_labels_per_s",
		        "infer_vit_b32_images_per_s", "infer_vit_b32_mfma_frac", "infer_vit_b32_b1024_mfma_frac", "infer_e2e_greedy_coalesced_labels_per_s", "infer_e2e_greedy_from_host_u8_labels_per_s")
		summary = {k: result.get(k) for k in keys if k in result}
		summary["roofline_frac"] = (result.get("roofline") or {}).get("frac")
		summary["shader_mhz_median"] = clock_info.get("mhz_median")
		note("summary " + json.dumps(summary))
	if world > 1:
		dist.barrier()
		dist.destroy_process_group()


def source_fingerprint() -> str:
	"""sha256[:16] over the sources whose change can move HBM traffic: every kernel source, the decoder / training host code, this file.  tools/collect_profile.sh
	stores it beside the PMC passes, tools/summarize_profile.py writes it into profiles/roofline_traffic.json, and `_profile_traffic` compares it with the tree that is
	running: a traffic figure profiled on other code is reported as stale, never as this run's."""
	import glob
	import hashlib
	h = hashlib.sha256()
	files = sorted(glob.glob(os.path.join(ROOT, "novic_amd", "csrc", "*.h*")) + glob.glob(os.path.join(ROOT, "novic_amd", "csrc", "*.cpp")) +
	               [os.path.join(ROOT, "novic_amd", f) for f in ("embedding_decoder.py", "train.py", "ops.py")] + [os.path.join(ROOT, "include", "novic_hip.h"), os.path.abspath(__file__)])
	for path in files:
		h.update(os.path.relpath(path, ROOT).encode())
		with open(path, "rb") as f:
			h.update(f.read())
	return h.hexdigest()[:16]


def _profile_traffic(key):
	"""(bytes, note): HBM bytes per launch / per step from the committed PMC passes of this same command (tools/collect_profile.sh -> profiles/roofline_traffic.json).
	The file names the source fingerprint it was profiled on; when that is not the tree running now, bytes is None and the note carries the stale figure."""
	try:
		with open(os.path.join(ROOT, "profiles", "roofline_traffic.json")) as f:
			rt = json.load(f)
	except (OSError, ValueError):
		return None, "no profiles/roofline_traffic.json"
	val, prof = rt.get(key), rt.get("source_sha16")
	if val is None:
		return None, "not profiled"
	if prof != source_fingerprint():
		return None, f"stale: {val} bytes were profiled on source {prof} ({rt.get('tag')}), this tree is {source_fingerprint()} -- rerun tools/collect_profile.sh"
	return val, f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on source {prof} ({rt.get('tag')})"


def gemm_class_roofline(model, spec, events, packed_rows, logit_rows, ms_per_step, n_steps):
	"""The dominant kernel of the step by time share: gemm256p_kernel<STORE_BF16> (csrc/gemm256.hip: 256 x 256 tiles, 8-phase K loop), which runs the step's large
	K-contiguous GEMMs -- per layer QKV [rows x 1536 x 512], the in-projection input gradient [rows x 512 x 1536] and the out-projection input gradient [rows x 512 x 512],
	once per step the logits GEMM [rows' x 6912 x 512], its input gradient [rows' x 512 x 6912] and the prefix MLP [8192 x 2048 x 512] (rows = the packed sequence positions,
	rows' = the output positions that count): the 21 dispatches per step of the rocprof summaries.  Every launch is bracketed by HIP events on
	the stream it is launched on, inside every timed step.  `achieved` = the class's algorithmic FLOP per step (2 M N K per launch, M = the rows that exist -- the
	device-side row count; SURVEY 8d's per-position terms x those rows) / the class's time per step, i.e. the time-weighted mean over its launches; `avg_us` = the mean
	launch duration; per shape: mean duration and fraction of the MFMA peak.  `algorithmic_bytes` = 2 (M K + N K + M N) summed over the launches of a step."""
	by = {}
	for name, M, N, K, t0, t1 in events:
		by.setdefault(name, dict(N=N, K=K, us=[]))["us"].append(1000.0 * t0.elapsed_time(t1))
	rows_of = {"qkv": float(packed_rows), "in_proj_dx": float(packed_rows), "out_proj_dx": float(packed_rows), "logits": float(logit_rows), "logits_dx": float(logit_rows),
	           "prefix_mlp": float(MICRO_B * ACCUM)}
	flop = us = nbytes = 0.0
	n_launch = 0
	per = {}
	for name, d in by.items():
		m = rows_of.get(name, 0.0)
		per_step = len(d["us"]) / float(n_steps)
		mean = sum(d["us"]) / len(d["us"])
		f = 2.0 * m * d["N"] * d["K"]
		flop += f * per_step
		us += mean * per_step
		nbytes += 2.0 * (m * d["K"] + d["N"] * d["K"] + m * d["N"]) * per_step
		n_launch += len(d["us"])
		per[name] = {"shape": [int(round(m)), d["N"], d["K"]], "launches_per_step": round(per_step, 2), "avg_us": round(mean, 2),
		             "mfma_frac": round(f / (mean * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)}
	ach = flop / (us * 1e-6) / 1e12 if us > 0 else 0.0
	traffic, note = _profile_traffic("gemm256_class_hbm_bytes_per_step")
	return {"kernel": "gemm256p_kernel<STORE_BF16>: the step's large K-contiguous GEMMs on 256 x 256 tiles (QKV, in-projection dX, out-projection dX x layers; logits, logits dX, prefix MLP)",
	        "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
	        "avg_us": round(us * n_steps / max(1, n_launch), 2), "launches_timed": n_launch, "traffic": traffic, "traffic_source": note, "traffic_unit": "bytes per optimizer step (all launches of the class)",
	        "algorithmic_bytes": int(nbytes), "algorithmic_flop_per_step": int(flop), "class_us_per_step": round(us, 1), "class_share_of_step": round(us / (1000.0 * ms_per_step), 4),
	        "per_shape": per}


def wgrad_roofline(model, spec, events, packed_rows, ms_per_step, n_steps, logit_rows=None):
	"""The dominant kernel of the step BY TIME SHARE: the weight-gradient class (dW = dY^T X with K = every sequence position of the step), and in it the
	self-attention in-projection gradient [3E x E] -- wgrad256_kernel<8> (256 x 256 tiles, split over K, raw partial sums to a workspace) followed by
	wgrad_reduce_kernel<8> (fixed-order sum + accumulate into the fp32 gradient).  HIP events bracket that launch PAIR on the stream it is launched on,
	inside every timed step; FLOP and bytes are counted for the rows that exist in the packed layout (`packed_rows`, averaged over the timed batches).
	`class_us_per_step` adds the other shapes served by the same kernel (out-projection, logits) for the time-share statement."""
	E = spec.hidden_dim
	by_shape = {}
	for name, m, n, t0, t1 in events:
		by_shape.setdefault((m, n), []).append(t0.elapsed_time(t1))
	two = (8 * E, E) in by_shape     # round 6: the pairs of TWO layers in one launch (novic_wgradn_bf16): priced as one [8E x E] gradient, the same FLOP
	paired = two or (4 * E, E) in by_shape  # the in-projection and out-projection gradients of a layer as one launch pair (novic_wgrad2_bf16): priced together
	dom = by_shape.get((8 * E, E) if two else ((4 * E, E) if paired else (3 * E, E)), [])
	mdom = 8 * E if two else (4 * E if paired else 3 * E)
	ms = sum(dom) / max(1, len(dom))
	per_step = max(1.0, len(dom) / float(n_steps * int(model.num_layers) / (2 if two else 1)))  # backward passes per optimizer step (1 when the micro-batches are merged)
	K = float(packed_rows) / per_step
	flops = 2.0 * K * mdom * E
	ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
	class_us = 1000.0 * sum(sum(v) for v in by_shape.values()) / n_steps
	what = ("in-projection + out-projection weight gradients of TWO layers in one launch pair (32 tiles x 8 parts): dW[3E x E] = dQKV^T LN1(x), dW[E x E] = g^T att, twice" if two else
	        "in-projection + out-projection weight gradients of a layer in one launch pair: dW[3E x E] = dQKV^T LN1(x), dW[E x E] = g^T att" if paired else
	        "in-projection weight gradient dW[3E x E] = dQKV^T LN1(x)")
	return {"kernel": "wgrad256p_kernel<8> + wgrad_reduce_kernel<8>: " + what, "shape": [mdom, E, int(round(K))], "bound": "mfma",
	        "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "avg_us": round(ms * 1000, 2),
	        "launches_timed": len(dom), "traffic": _profile_traffic("wgrad_in_proj_hbm_bytes_per_launch")[0], "traffic_source": _profile_traffic("wgrad_in_proj_hbm_bytes_per_launch")[1],
	        "algorithmic_bytes": int((2 if two else 1) * (2 * K * (3 * E + E) + (2 * K * (E + E) if paired else 0)) + 8 * mdom * E),
	        "class": "weight gradients on the 256-wide split-K kernel (attention pair x layers, feed-forward pair x layers, logits)",
	        "class_us_per_step": round(class_us, 1), "class_share_of_step": round(class_us / (1000.0 * ms_per_step), 4),
	        "per_shape_avg_us": {f"{m}x{n}": round(1000 * sum(v) / len(v), 2) for (m, n), v in sorted(by_shape.items())},
	        # the same launches as fractions of the MFMA peak: K = the packed rows for the layer weights, the output positions that count for the logits layer
	        "per_shape_mfma_frac": {f"{m}x{n}": round(2.0 * (float(logit_rows) if (m == spec.vocab_size and logit_rows) else K) * m * n / (sum(v) / len(v) * 1e-3) / 1e12
	                                                  / MFMA_BF16_PEAK_TFLOPS, 4) for (m, n), v in sorted(by_shape.items())}}


def measure_roofline(model, spec, device, ops, logits_events, rows_computed):
	"""Average duration of the dominant kernel -- the MFMA GEMM -- on its largest launch of the step (logits: [rows, 512] x [6912, 512]^T):
	HIP events recorded around that launch inside every TIMED training step, on the stream the kernel is launched on (torch's current stream).
	rows = the output positions the step computes logits for: the non-padded ones (`rows_computed`, averaged over the timed steps' batches) of the
	accum*512*7 the batch has -- FLOP and bytes are counted for THOSE rows only.
	`isolated_us` is the same GEMM (same row count) launched 20 times back to back after the steps (no neighbours, operands already in cache state)."""
	R_all, E, V = MICRO_B * ACCUM * (MAX_CONTENT + 1), spec.hidden_dim, spec.vocab_size
	R = int(round(rows_computed))
	ms = sum(s.elapsed_time(e) for s, e in logits_events) / max(1, len(logits_events))
	a = (torch.randn(R_all, E, device=device) * 0.5).to(torch.bfloat16)
	w = model._w16("logits_linear.weight")
	out = torch.empty(R_all, (V + 7) // 8 * 8, dtype=torch.bfloat16, device=device)
	for _ in range(3):
		ops.gemm(a, w, R, V, E, out=out)
	n = 20
	start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
	torch.cuda.synchronize()
	start.record()
	for _ in range(n):
		ops.gemm(a, w, R, V, E, out=out)
	stop.record()
	torch.cuda.synchronize()
	isolated_ms = start.elapsed_time(stop) / n
	flops = 2.0 * R * V * E
	ach = flops / (ms * 1e-3) / 1e12
	traffic, traffic_note = _profile_traffic("hbm_bytes_per_launch")
	return {"kernel": "gemm256p_kernel<STORE_BF16> logits GEMM", "shape": [R, V, E], "rows_allocated": R_all, "bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
	        "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4), "avg_us": round(ms * 1000, 2), "launches_timed": len(logits_events),
	        "isolated_us": round(isolated_ms * 1000, 2), "traffic": traffic, "traffic_source": traffic_note,
	        "algorithmic_bytes": 2 * (R * E + V * E + R * V)}


def measure_train_loop(device, accum, bare_value, legs=("resident", "streaming"), steps_per_chunk=8, chunks=5):
	"""The interface the reference exposes for training: `action_train` (train.py:977-1190) -- cache file -> loader -> GradAccum -> noise -> model -> optimizer -> schedule ->
	`training_loop`, which logs samples ("noun") per second per chunk (:1337).  A cache of the bench's shape is written first (RandomCacheWriter's recipe for the vectors,
	embedding_cache_writers.py:43, plus a synthetic noun vocabulary so that it has targets: 16 384 nouns of 1-6 tokens over 6 911 token strings + END -> V = 6 912), then the
	train action runs 1 + 4 chunks of 8 optimizer steps (8 192 samples each) over it: once with the cache resident in HBM (DeviceLoader's default) and once STREAMING
	(hbm budget forced to 0: embedding rows through pinned staging buffers and a copy stream).  Reported: the loop's own per-chunk rate (median of the chunks after the
	first) next to the bare train_step figure of this run."""
	import shutil
	import tempfile
	from novic_amd import embedders, embedding_cache, train as T
	tmp = tempfile.mkdtemp(prefix="novic_bench_")
	out = {}
	try:
		g = torch.Generator().manual_seed(2024)
		toks = [f"t{i}" for i in range(VOCAB - 1)]  # + END = V: 6 912, the bare step's vocabulary (until round 5 this wrote VOCAB - 3 strings -> V = 6 910 -- see `_Vs` in embedding_decoder.py for what that uncovered)
		spec_path = os.path.join(tmp, "embedder.json")
		with open(spec_path, "w") as f:
			json.dump(dict(tokens=toks, embed_dim=F_DIM), f)
		emb = embedders.Embedder.create(f"local:{spec_path}", device="cpu", load_model=False)
		n_nouns = 16384
		lens = torch.randint(1, MAX_CONTENT + 1, (n_nouns,), generator=g)
		ids = torch.randint(0, len(toks), (n_nouns, MAX_CONTENT), generator=g)
		cover = [" ".join(toks[i:i + 3]) for i in range(0, len(toks), 3)]  # every token string occurs, so the compact vocabulary is all of them: V = 6 912 exactly
		nouns = list(dict.fromkeys(cover + [" ".join(toks[int(t)] for t in row[:int(ln)]) for row, ln in zip(ids, lens)]))
		emb.configure_target(emb.create_target_config(nouns, with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True), nouns)
		n_embed = MICRO_B * accum * steps_per_chunk * 4
		cache_path = os.path.join(tmp, "bench_cache.bin")
		with embedding_cache.EmbeddingCacheWriter(cache_path, emb, n_embed, shuffle=False, use_targets=True, full_targets=True, target_nouns=nouns, num_embed_targets=1, default_weights=True,
		                                          embedder_strict=False) as w:
			left = n_embed
			while left > 0:
				b = min(65536, left)
				vec = torch.nn.functional.normalize(torch.randn(b, F_DIM, generator=g), dim=-1)
				w.write(embeds=vec, embed_targets=torch.randint(1, len(nouns) + 1, (b, 1), generator=g, dtype=torch.int32))
				left -= b
		chunk_scale = MICRO_B * accum * steps_per_chunk / len(nouns)
		for name, budget in (("train_loop", None), ("train_loop_streaming", "0")):
			if ("streaming" if budget else "resident") not in legs:
				continue
			if budget is None:
				os.environ.pop("NOVIC_LOADER_HBM_BUDGET", None)
			else:
				os.environ["NOVIC_LOADER_HBM_BUDGET"] = budget
			cfg = T.default_train_config(embedder_spec=f"local:{spec_path}", embedding_dataset=cache_path, strict_embedder=False, batch_size=MICRO_B, accum_factor=accum,
			                             chunk_scale=chunk_scale, max_chunks=chunks, max_epochs=0, noise_scheme="GaussElemUniformAngle", noise_vec_norm=3.25, noise_angle_min=45.0,
			                             noise_angle_max=75.0, noise_mix_ratio=0.15, save_every_min=10 ** 6, save_every_max=10 ** 6, save_top1_min=100.0, determ=True, determ_seed=7)
			rates = []
			res = T.action_train(cfg, os.path.join(tmp, name), False, log=lambda m: None, on_chunk=lambda info: rates.append(float(info["samples_per_s"])))
			torch.cuda.synchronize()
			steady = sorted(rates[1:])
			out[f"{name}_samples_per_s"] = round(steady[len(steady) // 2], 1)
			out[f"{name}_vs_bare_step"] = round(steady[len(steady) // 2] / bare_value, 4)
			del res
		os.environ.pop("NOVIC_LOADER_HBM_BUDGET", None)
		out["train_loop_config"] = (f"action_train on a {n_embed}-embedding cache written here ({len(nouns)} nouns, F {F_DIM}, V {VOCAB}), batch {MICRO_B} x accum {accum}, {chunks} chunks of "
		                            f"{steps_per_chunk} optimizer steps; per-chunk rate as training_loop logs it (median of chunks 2..{chunks}; the chunk's one host synchronisation is inside)")
	finally:
		shutil.rmtree(tmp, ignore_errors=True)
	return out


def measure_multiset(device, rank, world, dist, accum, steps=5, persistent_cus=None):
	"""configs[4]: the multiset fine-tune step -- ViT-H/14 embedding width F = 1024, M = 3 targets per embedding with descending weights that sum to 1
	(embedding_dataset.py:20), so 3 x 512 = 1536 sequences per micro-batch -- same decoder, noise, optimizer and accumulation as the headline step.
	Samples = embeddings (each carries its M targets).  Barrier / max-over-ranks timing as for the headline number."""
	from novic_amd import train as T, embedding_noise
	M, F = 3, 1024
	spec = WorkloadSpec(embed_dim=F, vocab_size=VOCAB, token_length=CMAX)
	torch.manual_seed(0)
	model = build_decoder(spec, dropout=0.1, device=device, multi_length=M)
	dp = T.DataParallel(persistent_cus=persistent_cus)
	dp.broadcast_parameters(model.flat_parameters())
	model.train()
	opt = T.FusedAdamW(model, lr=1.5e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	noise = embedding_noise.EmbeddingNoise.create("GaussElemUniformAngle", F, 3.25, 45.0, 75.0, 0.0, 0.15)

	def micro(seed):
		g = torch.Generator().manual_seed(seed)
		embed = torch.nn.functional.normalize(torch.randn(MICRO_B, F, generator=g), dim=-1)
		lens = torch.randint(1, MAX_CONTENT + 1, (MICRO_B, M), generator=g)
		col = torch.arange(MAX_CONTENT + 1).view(1, 1, -1)
		target = torch.randint(1, VOCAB, (MICRO_B, M, MAX_CONTENT + 1), generator=g) * (col < lens.unsqueeze(-1))
		w = torch.rand(MICRO_B, M, generator=g).sort(dim=1, descending=True)[0]
		return embed.to(device), target.to(device), (col > lens.unsqueeze(-1)).to(device), (w / w.sum(dim=1, keepdim=True)).to(device)

	pool = [micro(777 + rank * 1000 + j) for j in range(accum)]
	embeds = torch.stack([mb[0] for mb in pool])

	def one_step():
		fresh = embeds.clone()
		return T.train_step(model, opt, [(fresh[j], t, m, w) for j, (_, t, m, w) in enumerate(pool)], embed_noise=noise, dp=dp)

	for _ in range(2):
		one_step()
	torch.cuda.synchronize()
	if world > 1:
		dist.barrier()
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(steps):
		stats, gnorm = one_step()
	torch.cuda.synchronize()
	if world > 1:
		dist.barrier()
	torch.cuda.synchronize()
	dt = time.perf_counter() - t0
	if world > 1:
		t = torch.tensor([dt], dtype=torch.float64, device=device)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		dt = float(t)
	loss = float((stats[1] / stats[0]).mean())
	assert math.isfinite(loss) and math.isfinite(float(gnorm))
	model._ws.clear()
	return {"train_multiset_samples_per_s": round(MICRO_B * accum * world * steps / dt, 1), "train_multiset_ms_per_step": round(1000 * dt / steps, 3),
	        "train_multiset_config": {"embed_dim": F, "targets_per_sample": M, "sequences_per_step": MICRO_B * M * accum, "weights": "descending, sum 1", "loss_last": round(loss, 4)}}


def measure_decode(spec, device, B, world, dist):
	"""Decoder-only labels/s from random unit embeddings: greedy and beam-4, generation length pinned to G = Cmax-1 by zeroing the END row
	of the tied embedding.  Per GPU batch B, no collective."""
	torch.manual_seed(1)
	model = build_decoder(spec, dropout=0.0, device=device)
	with torch.no_grad():
		model.logits_linear.weight[0].zero_()
	model.eval()
	g = torch.Generator().manual_seed(99)
	embed = torch.nn.functional.normalize(torch.randn(B, spec.embed_dim, generator=g), dim=-1).to(device)
	out = {}
	# a synthetic noun vocabulary of the released size (42 919 nouns, infer.py:174) for the default guided configuration beam_k10_vnone_gp_t1_a0
	W, G = 42919, spec.token_length - 1
	lens = torch.randint(1, 5, (W,), generator=g)
	nouns = torch.randint(1, spec.vocab_size, (W, spec.token_length), generator=g) * (torch.arange(spec.token_length).unsqueeze(0) < lens.unsqueeze(1))
	nouns = torch.unique(nouns, dim=0).to(device)
	for name, fn in (("greedy", lambda: model.generate(embed, False, True, 1.0, 0.0, None, None, False)),
	                 ("beam4", lambda: model.generate_beam(embed, 4, 1.0, 0.0, None, False, 0.0, None, False)),
	                 ("beam10_guided", lambda: model.generate_beam(embed, 10, 1.0, 0.0, None, False, 0.0, nouns, False))):
		with torch.no_grad():
			for _ in range(3):  # call 1 eager, call 2 captures the step graph, call 3 replays it
				fn()
			torch.cuda.synchronize()
			reps = 10
			t0 = time.perf_counter()
			for _ in range(reps):
				res = fn()
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / reps
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		steps = res[0].shape[-1]
		out[f"infer_{name}_labels_per_s"] = round(B * world / dt, 1)
		out[f"infer_{name}_steps"] = int(steps)
	# two / three independent batches of B decoded concurrently (each lane needs a hardware queue of its own: GPU_MAX_HW_QUEUES = 8 above; four lanes oversubscribe them) (generate_many: one stream + session per batch; outputs bit-identical to one-at-a-time decoding,
	# tests/test_gpu_fullsize_properties.py): the throughput form of the same launches -- a decode step at 256 rows leaves most CUs idle
	for lanes in ((2, 3) if world == 1 else ()):  # (N > 1: GPU_MAX_HW_QUEUES stays at the runtime's default beside RCCL, where three lanes share queues)
		es = [torch.nn.functional.normalize(torch.randn(B, spec.embed_dim, generator=g), dim=-1).to(device) for _ in range(lanes)]
		for name, fn in (("greedy", lambda: model.generate_many(es, False, True, 1.0, 0.0, None, None, False)),
		                 ("beam4", lambda: model.generate_beam_many(es, 4, 1.0, 0.0, None, False, 0.0, None, False)),
		                 ("beam10_guided", lambda: model.generate_beam_many(es, 10, 1.0, 0.0, None, False, 0.0, nouns, False))):
			with torch.no_grad():
				for _ in range(3):
					fn()
				torch.cuda.synchronize()
				reps = 8
				t0 = time.perf_counter()
				for _ in range(reps):
					fn()
				torch.cuda.synchronize()
				dt = (time.perf_counter() - t0) / reps
			if dist is not None:
				t = torch.tensor([dt], dtype=torch.float64, device=device)
				dist.all_reduce(t, op=dist.ReduceOp.MAX)
				dt = float(t)
			out[f"infer_{name}_{lanes}x{B}_concurrent_labels_per_s"] = round(lanes * B * world / dt, 1)
	# the same greedy / beam-4 decode at four times the batch (how the latency-bound B = 256 figure scales with rows per step)
	big = torch.nn.functional.normalize(torch.randn(4 * B, spec.embed_dim, generator=g), dim=-1).to(device)
	for name, fn in ((f"greedy_b{4 * B}", lambda: model.generate(big, False, True, 1.0, 0.0, None, None, False)),
	                 (f"beam4_b{4 * B}", lambda: model.generate_beam(big, 4, 1.0, 0.0, None, False, 0.0, None, False))):
		with torch.no_grad():
			for _ in range(3):
				fn()
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(5):
				fn()
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / 5
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_labels_per_s"] = round(4 * B * world / dt, 1)
	# image path: random-pixel 224x224 batches through the native ViT-B/32 tower (random init), alone and followed by greedy / beam-4 decoding
	from novic_amd import clip_vit
	vit = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=3).to(device)
	images = torch.randn(B, 3, 224, 224, generator=g).to(device)
	# The metric's tower is openai:ViT-B/32, which the reference runs as clip's HALF-PRECISION model (embedders.py:488-489): every ViT-B/32 leg below runs with the residual
	# stream in IEEE half, as local_clip.OpenAIEmbedder sets it (NativeViT.half_stream, round 6).  The fp32-stream form of the same tower (what rounds 1-5 measured, and
	# what an open_clip ViT-B/32 under autocast would run) is timed first, alone, for the record.
	for nb, imgs in ((B, images), (4 * B, torch.randn(4 * B, 3, 224, 224, generator=g).to(device))):
		with torch.no_grad():
			for _ in range(3):
				vit(imgs)
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(8):
				vit(imgs)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / 8
		out[f"infer_vit_b32_fp32_stream{'' if nb == B else f'_b{nb}'}_images_per_s"] = round(nb * world / dt, 1)
	del imgs
	vit._rt_reset()
	vit.half_stream = True
	pipelines = (("vit_b32_images", lambda: vit(images)),
	             ("e2e_greedy_labels", lambda: model.generate(vit(images), False, True, 1.0, 0.0, None, None, False)),
	             ("e2e_beam4_labels", lambda: model.generate_beam(vit(images), 4, 1.0, 0.0, None, False, 0.0, None, False)))
	for name, fn in pipelines:
		with torch.no_grad():
			for _ in range(3):
				fn()
			torch.cuda.synchronize()
			reps = 10
			t0 = time.perf_counter()
			for _ in range(reps):
				fn()
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / reps
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_per_s"] = round(B * world / dt, 1)
	# the same two pipelines over a SEQUENCE of image batches, pipelined (embedders.pipeline_image_batches, what NOVICModel.classify_image_batches runs): the tower of the
	# next batch on a stream of its own, its persistent GEMM grids on embedders.pipeline_budget(rows) of the 256 CUs (184 at batch 256, 208 at 1 024), beside the decoding of the current one; throughput form of the two lines above
	from novic_amd import embedders
	seq = [images] + [torch.randn(B, 3, 224, 224, generator=g).to(device) for _ in range(3)]
	for name, dec in (("e2e_greedy_pipelined_labels", lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)),
	                  ("e2e_beam4_pipelined_labels", lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False))):
		with torch.no_grad():
			for _ in range(2):
				for e in embedders.pipeline_image_batches(vit, seq, device):
					dec(e)
			torch.cuda.synchronize()
			reps = 4
			t0 = time.perf_counter()
			for e in embedders.pipeline_image_batches(vit, seq * reps, device):
				dec(e)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / (reps * len(seq))
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_per_s"] = round(B * world / dt, 1)
	# ... and with COALESCED towers (round 5; what Embedder.inference_image_batches / NOVICModel.classify_image_batches do by default at this size): four consecutive caller
	# batches of B images run as one tower forward over 4 B images (600 instead of 150 tiles in the out-projection / fc2 GEMMs), the embeddings still handed out -- and decoded -- per
	# caller batch of B.  Bit-identical embeddings and labels (tests/test_gpu_fullsize_properties.py).
	from novic_amd.infer import split_decode_groups, NOVICModel

	def run_coalesced(src, dec, rows, n=4, lanes=1, dec_many=None):
		"""What NOVICModel.classify_image_batches does: one tower launch per n = 4 caller batches, <= `rows` rows of it per decode call, and (round 6) `lanes` = 2 decode
		calls at the same time on lanes of their own (dec_many: generate_many / generate_beam_many) with the tower's workgroup budget that goes with it."""
		held = []
		def budget(arg):  # (as Embedder.inference_image_batches: host batches keep the reservation -- their H2D copies run beside the tower too)
			first = arg[0] if isinstance(arg, (list, tuple)) else arg
			return embedders.pipeline_budget(sum(t.shape[0] for t in arg) * 50 if isinstance(arg, (list, tuple)) else arg.shape[0] * 50, lanes if first.device.type != "cpu" else 1)
		for e, sizes in embedders.pipeline_image_batches(vit, src, device, budget, coalesce=n, grouped=True):
			for (a, b), _ in split_decode_groups(sizes, rows):
				if lanes == 1 or src[0].device.type == "cpu":  # (batches staged from the host: one decode call at a time, as NOVICModel.classify_image_batches does)
					dec(e[a:b])
					continue
				held.append(e[a:b])
				if len(held) == lanes:
					dec_many(held)
					held = []
		for h in held:
			dec(h)
	out["infer_coalesce"] = {"tower_batches_per_launch": 4, "decode_rows_per_call": NOVICModel.decode_rows,
	                         "note": "caller batches stay at batch_per_gpu images; one tower launch per 4 of them, decoded <= decode_rows_per_call rows per call, results handed out per "
	                                 "caller batch: embeddings, ids, scores bit-identical to one call per batch (tests/test_gpu_fullsize_properties.py); *_coalesced512_* keys: two "
	                                 "decode calls per tower launch"}
	g_one, b_one = (lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)), (lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False))
	g_many, b_many = (lambda es: model.generate_many(es, False, True, 1.0, 0.0, None, None, False)), (lambda es: model.generate_beam_many(es, 4, 1.0, 0.0, None, False, 0.0, None, False))
	out["infer_coalesce"]["decode_lanes"] = NOVICModel.decode_lanes
	out["infer_coalesce"]["note_lanes"] = ("*_coalesced_* keys: as NOVICModel.classify_image_batches runs by default since round 6 -- the embeddings of TWO tower launches decoded at the same "
	                                       "time on lanes of their own (generate_many: bit-identical to one call each), the tower on all 256 CUs; *_coalesced_one_call_*: one decode call "
	                                       "at a time beside a tower on 208 CUs, the form of rounds 4-5 and what batches staged from the host still get (*_from_host_*: a second lane loses there, "
	                                       "NOVICModel.decode_lanes); 48 caller batches per timed run (the pipeline's fill and drain are a fixed cost)")
	for name, dec, rows, lanes, many in (("e2e_greedy_coalesced_labels", g_one, NOVICModel.decode_rows, NOVICModel.decode_lanes, g_many),
	                                     ("e2e_beam4_coalesced_labels", b_one, NOVICModel.decode_rows, NOVICModel.decode_lanes, b_many),
	                                     ("e2e_greedy_coalesced_one_call_labels", g_one, NOVICModel.decode_rows, 1, None),
	                                     ("e2e_beam4_coalesced_one_call_labels", b_one, NOVICModel.decode_rows, 1, None),
	                                     ("e2e_greedy_coalesced512_labels", g_one, 2 * B, 1, None)):
		with torch.no_grad():
			for _ in range(3):
				run_coalesced(seq * 2, dec, rows, lanes=lanes, dec_many=many)
			torch.cuda.synchronize()
			reps = 12
			t0 = time.perf_counter()
			run_coalesced(seq * reps, dec, rows, lanes=lanes, dec_many=many)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / (reps * len(seq))
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_per_s"] = round(B * world / dt, 1)
	# EIGHT caller batches per tower launch and decode call (not the default: twice the look-ahead and staging memory of four -- `Embedder.coalesce_max`): for the record
	with torch.no_grad():
		for _ in range(2):
			run_coalesced(seq * 4, g_one, 8 * B, n=8, lanes=NOVICModel.decode_lanes, dec_many=g_many)
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		run_coalesced(seq * 16, g_one, 8 * B, n=8, lanes=NOVICModel.decode_lanes, dec_many=g_many)
		torch.cuda.synchronize()
		dt = (time.perf_counter() - t0) / (16 * len(seq))
	if dist is not None:
		t = torch.tensor([dt], dtype=torch.float64, device=device)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		dt = float(t)
	out["infer_e2e_greedy_coalesced8_labels_per_s"] = round(B * world / dt, 1)
	# the same tower and pipeline at FOUR times the batch (the reference uses one batch size for tower and decoder, infer.py:99-101; its default is 128, nothing fixes it): the
	# tower's single-round GEMMs fill the chip (150 -> 600 tiles) and a decode step carries four times the rows per launch
	big_seq = [torch.randn(4 * B, 3, 224, 224, generator=g).to(device) for _ in range(3)]
	with torch.no_grad():
		for _ in range(3):
			vit(big_seq[0])
		torch.cuda.synchronize()
		t0 = time.perf_counter()
		for _ in range(5):
			vit(big_seq[0])
		torch.cuda.synchronize()
		dt = (time.perf_counter() - t0) / 5
	if dist is not None:
		t = torch.tensor([dt], dtype=torch.float64, device=device)
		dist.all_reduce(t, op=dist.ReduceOp.MAX)
		dt = float(t)
	out[f"infer_vit_b32_b{4 * B}_images_per_s"] = round(4 * B * world / dt, 1)
	for name, dec in ((f"e2e_greedy_b{4 * B}_pipelined_labels", lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)),
	                  (f"e2e_beam4_b{4 * B}_pipelined_labels", lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False))):
		with torch.no_grad():
			for _ in range(2):
				for e in embedders.pipeline_image_batches(vit, big_seq, device):
					dec(e)
			torch.cuda.synchronize()
			reps = 3
			t0 = time.perf_counter()
			for e in embedders.pipeline_image_batches(vit, big_seq * reps, device):
				dec(e)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / (reps * len(big_seq))
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_per_s"] = round(4 * B * world / dt, 1)
	del big_seq
	# ... and from the HOST, as the reference's interface has it (`inference_image` takes CPU images, embedders.py:759-764): the same batches as CPU fp32 tensors -- pinned, as
	# a DataLoader with pin_memory=True delivers them (classification_dataset.py:220), and pageable -- through embedders.ImageStager: pre-pinned staging ring, copy stream, the
	# H2D copy of batch i + 2 under the tower of batch i + 1 and the decoding of batch i.  154 MB per batch over PCIe: the link rate bounds these figures, not the GPU.
	host_pageable = [im.cpu() for im in seq]
	host_pinned = [im.pin_memory() for im in host_pageable]
	stager = embedders.image_stager(device)
	legs = (("e2e_greedy_from_host_labels", host_pinned, lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)),
	        ("e2e_beam4_from_host_labels", host_pinned, lambda e: model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False)),
	        ("e2e_greedy_from_pageable_host_labels", host_pageable, lambda e: model.generate(e, False, True, 1.0, 0.0, None, None, False)))
	for name, src, dec in legs:
		with torch.no_grad():
			for _ in range(2):
				for e in embedders.pipeline_image_batches(vit, src, device):
					dec(e)
			torch.cuda.synchronize()
			reps = 3
			t0 = time.perf_counter()
			for e in embedders.pipeline_image_batches(vit, src * reps, device):
				dec(e)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / (reps * len(src))
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_per_s"] = round(B * world / dt, 1)
	# from the host with coalesced towers, as normalised fp32 images (the reference's interface) and as uint8 pixels (Embedder.get_image_transform(uint8=True): ToTensor /
	# Normalize applied by the tower's first kernel, same fp32 arithmetic, bit-identical embeddings; 38.5 instead of 154 MB per batch over PCIe)
	mean_t, std_t = (torch.tensor(v).view(1, 3, 1, 1) for v in vit._pixel_norm())
	host_u8 = [torch.randint(0, 256, (B, 3, 224, 224), generator=g, dtype=torch.uint8).pin_memory() for _ in range(len(seq))]
	legs = (("e2e_greedy_from_host_coalesced_labels", host_pinned, g_one, g_many),
	        ("e2e_greedy_from_host_u8_labels", host_u8, g_one, g_many),
	        ("e2e_beam4_from_host_u8_labels", host_u8, b_one, b_many))
	for name, src, dec, many in legs:  # (host batches: run_coalesced decodes them one call at a time whatever `lanes` says, as the product does)
		with torch.no_grad():
			for _ in range(3):
				run_coalesced(src * 2, dec, NOVICModel.decode_rows, lanes=NOVICModel.decode_lanes, dec_many=many)
			torch.cuda.synchronize()
			reps = 12
			t0 = time.perf_counter()
			run_coalesced(src * reps, dec, NOVICModel.decode_rows, lanes=NOVICModel.decode_lanes, dec_many=many)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / (reps * len(src))
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_per_s"] = round(B * world / dt, 1)
	del host_u8
	# the H2D copies alone (pinned source, copy stream): what the link delivers for these batches
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for i in range(8):
		_, _, slot = stager.stage(host_pinned[i % len(host_pinned)])
		embedders.ImageStager.release(slot, stager.copy_stream)
	stager.copy_stream.synchronize()
	out["infer_h2d_GB_per_s"] = round(8 * host_pinned[0].numel() * 4 / (time.perf_counter() - t0) / 1e9, 2)
	out["infer_from_host_note"] = (f"{B} x 3 x 224 x 224 fp32 = {host_pinned[0].numel() * 4 / 1e6:.0f} MB per batch over PCIe; at the measured H2D rate that alone is "
	                               f"{host_pinned[0].numel() * 4 / 1e9 / out['infer_h2d_GB_per_s'] * 1e3:.2f} ms per batch = {B / (host_pinned[0].numel() * 4 / 1e9 / out['infer_h2d_GB_per_s']):.0f} images/s")
	del vit, seq, host_pageable, host_pinned
	# configs[3]: OpenCLIP ViT-L/14 image tower (F = 768, 257 tokens, width 1024, 24 layers: 162 GFLOP per image) + beam-4 decode through a decoder
	# built for F = 768, per-step hipGraphs -- random init, random pixels, same batch per GPU
	spec_l = dataclasses.replace(spec, embed_dim=clip_vit.VIT_L_14.embed_dim)
	model_l = build_decoder(spec_l, dropout=0.0, device=device)
	with torch.no_grad():
		model_l.logits_linear.weight[0].zero_()
	model_l.eval()
	vit_l = clip_vit.NativeViT(clip_vit.VIT_L_14, seed=5).to(device)
	for name, fn in (("vit_l14_images", lambda: vit_l(images)),
	                 ("l14_e2e_beam4_labels", lambda: model_l.generate_beam(vit_l(images), 4, 1.0, 0.0, None, False, 0.0, None, False))):
		with torch.no_grad():
			for _ in range(3):
				fn()
			torch.cuda.synchronize()
			reps = 5
			t0 = time.perf_counter()
			for _ in range(reps):
				fn()
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / reps
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{name}_per_s"] = round(B * world / dt, 1)
	out["infer_vit_l14_mfma_frac"] = round(out["infer_vit_l14_images_per_s"] / world * clip_vit.VIT_L_14.flops_per_image() / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
	del vit_l, model_l
	# the image towers of the released checkpoints (reference README.md:293-298: openclip:timm/ViT-B-16-SigLIP, timm/ViT-SO400M-14-SigLIP, apple/DFN5B-CLIP-ViT-H-14-378):
	# timm trunk + attention-pool head / CLIP ViT-H/14 at 378 pixels (730 tokens, heads of 80), random init; SO400M's 72-wide heads / 4304-wide MLP run zero-padded to
	# 80 / 4352 (the FLOP counted are the model's own, not the padded ones)
	from novic_amd import siglip
	h14 = clip_vit.ViTConfig(378, 14, 1280, 32, 16, 4.0, 1024, quick_gelu=True)
	images378 = None
	for key, scfg in (("siglip_b16", siglip.SigLIPVisionConfig(224, 16, 768, 12, 12, 3072)), ("siglip_so400m14", siglip.SigLIPVisionConfig(224, 14, 1152, 27, 16, 4304)),
	                  ("vit_h14_378", h14)):
		tower = (clip_vit.NativeViT(scfg, seed=6) if scfg is h14 else siglip.NativeSigLIPViT(scfg, seed=6)).to(device)
		if scfg is h14:
			images378 = torch.randn(B, 3, 378, 378, generator=g).to(device)
		x = images378 if scfg is h14 else images
		with torch.no_grad():
			for _ in range(2):
				tower(x)
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(3):
				tower(x)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / 3
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out[f"infer_{key}_images_per_s"] = round(B * world / dt, 1)
		out[f"infer_{key}_mfma_frac"] = round(B / dt * scfg.flops_per_image() / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
		del tower, x
	del images378
	# text tower (what fills the embedding cache the training step reads): 77-token CLIP rows, ViT-B/32 text dims, random init
	from novic_amd import clip_text
	txt = clip_text.NativeTextTower(clip_text.TEXT_B_32, seed=4).to(device)
	tids = torch.randint(1, 49406, (B, 77), generator=g)
	tids[:, 0], tids[:, -1] = 49406, 49407
	tids = tids.to(device)
	for half in (False, True):  # (the text tower of openai:ViT-B/32: the reference runs it in half precision too -- first the fp32-stream form for the record, then as OpenAIEmbedder sets it)
		txt.half_stream = half
		with torch.no_grad():
			for _ in range(3):
				txt(tids)
			torch.cuda.synchronize()
			t0 = time.perf_counter()
			for _ in range(10):
				txt(tids)
			torch.cuda.synchronize()
			dt = (time.perf_counter() - t0) / 10
		if dist is not None:
			t = torch.tensor([dt], dtype=torch.float64, device=device)
			dist.all_reduce(t, op=dist.ReduceOp.MAX)
			dt = float(t)
		out["infer_text_b32_texts_per_s" if half else "infer_text_b32_fp32_stream_texts_per_s"] = round(B * world / dt, 1)
	out["infer_text_b32_mfma_frac"] = round(out["infer_text_b32_texts_per_s"] / world * clip_text.TEXT_B_32.flops_per_text() / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
	fl = clip_vit.VIT_B_32.flops_per_image()
	out["infer_vit_b32_mfma_frac"] = round(out["infer_vit_b32_images_per_s"] / world * fl / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
	out[f"infer_vit_b32_b{4 * B}_mfma_frac"] = round(out[f"infer_vit_b32_b{4 * B}_images_per_s"] / world * fl / 1e12 / MFMA_BF16_PEAK_TFLOPS, 4)
	out["infer_config"] = {"batch_per_gpu": B, "decode_steps_forced": spec.token_length - 1, "decoder_only_embeddings": "random unit vectors",
	                       "beam10_guided": f"{nouns.shape[0]} synthetic nouns of 1-4 tokens, guided (gp), early exit when every beam has spelt a noun",
	                       "image_tower": "openai:ViT-B/32 224px random init, residual stream in IEEE half as the reference's fp16 clip model (the *_fp32_stream_* keys: the same tower with an fp32 stream), random-pixel images resident in HBM", "vit_flop_per_image": fl,
	                       "config4": "ViT-L/14 224px tower (F = 768) + beam-4 decode, random init", "vit_l14_flop_per_image": clip_vit.VIT_L_14.flops_per_image()}
	return out


def cpu_baseline(spec):
	"""The oracle port (plain PyTorch fp32 on the host cores; dropout-free, so faster than the reference's own CPU path) on a bounded sample:
	whole train steps of ONE 512-sample micro-batch each (noise-free forward + backward + clip + AdamW), ~10-30 s of CPU work."""
	from oracle import decoder_oracle as O  # the ONLY place the bench touches oracle/: the timed CPU baseline
	torch.set_num_threads(host_threads())
	spec = O.DecoderSpec(embed_dim=spec.embed_dim, vocab_size=spec.vocab_size, token_length=spec.token_length)
	sd = O.init_state_dict(spec, seed=0)
	params = {k: v.clone() for k, v in sd.items() if k != "causality_mask"}
	state = {}
	mb = tuple(None if t is None else t.cpu() for t in synth_micro_batch(spec, MICRO_B, 4321, "cpu"))

	def step(i):
		req = {k: v.clone().requires_grad_(True) for k, v in params.items()}
		total, _ = O.loss_for_step(dict(req, causality_mask=sd["causality_mask"]), spec, [mb])
		total.backward()
		O.clip_and_adamw(params, {k: v.grad for k, v in req.items()}, state, i, 1.5e-3)

	step(1)
	t0 = time.perf_counter()
	n = 0
	while n < 3 or (time.perf_counter() - t0 < 12 and n < 40):
		n += 1
		step(n + 1)
	dt = time.perf_counter() - t0
	try:
		visible = len(os.sched_getaffinity(0))
	except AttributeError:
		visible = os.cpu_count() or 1
	out = {"value": round(MICRO_B * n / dt, 1), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
	       "cores_note": f"a {torch.get_num_threads()}-thread share of the host ({visible} logical CPUs visible to this process): one GPU's share of the box, not every physical core",
	       "sample": f"{n} optimizer steps of one {MICRO_B}-sample micro-batch (fp32, no dropout/noise), {dt:.1f}s"}
	# the inference half of the metric on the same host cores: greedy / beam-4 labels/s with the generation length pinned to
	# G = Cmax-1 as on the GPU (END row of the tied embedding zeroed), and ViT-B/32 images/s through the oracle tower -- a few seconds each
	with torch.no_grad():
		sd_dec = dict(sd)
		sd_dec["logits_linear.weight"] = sd["logits_linear.weight"].clone()
		sd_dec["logits_linear.weight"][0].zero_()
		g = torch.Generator().manual_seed(99)
		Bc = 64
		embed = torch.nn.functional.normalize(torch.randn(Bc, spec.embed_dim, generator=g), dim=-1)
		for name, fn, per in (("greedy", lambda: O.generate(sd_dec, spec, embed), 1), ("beam4", lambda: O.generate_beam(sd_dec, spec, embed, 4), 1)):
			fn()
			t0 = time.perf_counter()
			n = 0
			while n < 2 or (time.perf_counter() - t0 < 4 and n < 20):
				res = fn()
				n += 1
			dt = time.perf_counter() - t0
			out[f"infer_{name}_labels_per_s"] = round(Bc * n / dt, 1)
			out[f"infer_{name}_steps"] = int(res[0].shape[-1])
		from oracle import vit_oracle as VO
		vspec = VO.ViTSpec(image_size=224, patch_size=32, width=768, layers=12, heads=12, embed_dim=512, quick_gelu=True)
		vsd = VO.init_state_dict(vspec, 3)
		images = torch.randn(16, 3, 224, 224, generator=g)
		VO.encode_image(vsd, vspec, images)
		t0 = time.perf_counter()
		n = 0
		while n < 2 or (time.perf_counter() - t0 < 4 and n < 20):
			VO.encode_image(vsd, vspec, images)
			n += 1
		dt = time.perf_counter() - t0
		out["infer_vit_b32_images_per_s"] = round(16 * n / dt, 1)
	out["infer_sample"] = f"decoder: batches of {Bc} random unit embeddings, {spec.token_length - 1} forced steps; ViT-B/32: batches of 16 random-pixel 224px images (fp32 oracle, random init)"
	return out


if __name__ == "__main__":
	main()
