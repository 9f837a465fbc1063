"""Decoder training on MI355X: fused optimizer, data-parallel gradient exchange and the training loop.

Mirrors the train slice of reference train.py: ``TrainLoopConfig`` / ``TrainLoopState`` (:927-974, same fields), the optimizer and
schedule set-up of ``action_train`` (:1103-1165: AdamW(beta 0.9/0.95), weight decay on >= 2-D params only, LinearLR warm-up and
CosineAnnealingLR stepped once per CHUNK), ``training_loop`` (:1193-1429: mean-shift, noise, loss = loss_sum / loss_basis / accum,
global-norm clip before the step, EWA loss / top-1 bookkeeping, chunk accounting, save policy) and ``save_train_checkpoint``
(:1433-1479, same dict layout so ``infer.NOVICModel`` of either side can read it).

What is different in HOW:
* one optimizer step = ONE merged forward/backward over all `accum` micro-batches (``PrefixedIterDecoder.forward_backward`` with
  ``group_rows`` = micro-batch size): the loss is still the mean of per-micro-batch means, the GEMMs are `accum` times taller;
* no per-batch ``.item()``: per-micro-batch (basis, loss_sum, correct, tokens) stay on the device and are replayed through the
  same EWA recursion once per chunk (one host sync per chunk instead of 4 per batch);
* data parallel (new; the reference is single-process): one process per GPU, every rank takes its own micro-batches, gradients are
  summed with ONE RCCL all-reduce of the flat fp32 gradient buffer per optimizer step (loss scale 1 / (accum * world)), the clip
  uses the post-reduce global norm, so every rank applies the identical update.
"""
from __future__ import annotations

import dataclasses
import datetime
import math
import os
import sys
import time
from typing import Any, Callable, Iterable, Optional, Sequence

import torch

from . import embedders, embedding_cache, embedding_dataset, embedding_decoder, embedding_noise, infer, ops, utils


# ------------------------------------------------------------------------------------------------------------------------------
# optimizer + schedules
# ------------------------------------------------------------------------------------------------------------------------------

class FusedAdamW:
	"""Clip-to-global-norm + decoupled AdamW over the decoder's flat parameter buffer in two launches (norm, update).

	Semantics of ``torch.nn.utils.clip_grad_norm_(max_norm)`` followed by ``torch.optim.AdamW.step()`` with weight decay applied to
	the >= 2-D parameters only (reference train.py:1103-1119, :1280-1286).
	"""

	def __init__(self, model: embedding_decoder.PrefixedIterDecoder, lr: float, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.1, max_norm: float = 1.0,
	             weight_decay_1d: bool = False):
		self.model = model
		self.param_groups = [dict(lr=lr, initial_lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)]
		self.max_norm = max_norm
		self.weight_decay_1d = weight_decay_1d
		self.step_count = 0
		flat = model.flat_parameters()
		self.exp_avg = torch.zeros_like(flat)
		self.exp_avg_sq = torch.zeros_like(flat)
		self.grad_norm = torch.zeros(1, dtype=torch.float32, device=flat.device)
		self._partial = torch.empty(1024, dtype=torch.float64, device=flat.device)

	@property
	def lr(self) -> float:
		return self.param_groups[0]["lr"]

	def zero_grad(self, set_to_none: bool = False):
		self.model.flat_grad().zero_()

	def step(self) -> torch.Tensor:
		"""Returns the (pre-clip) global gradient norm as a 1-element device tensor; no host sync."""
		g = self.param_groups[0]
		self.step_count += 1
		b1, b2 = g["betas"]
		flat, grad = self.model.flat_parameters(), self.model.flat_grad()
		ops.grad_norm(grad, self._partial, self.grad_norm)
		n_decay = flat.numel() if self.weight_decay_1d else self.model.num_decay_elements
		self.model.flat_shadow()
		# hyper-parameters travel BY VALUE in the kernel arguments: the host may run any number of steps ahead of the device (one sync per chunk)
		ops.adamw_step(flat, grad, self.exp_avg, self.exp_avg_sq, self.model._flat16, n_decay, self.grad_norm if self.max_norm > 0 else None, lr=g["lr"], beta1=b1, beta2=b2,
		               eps=g["eps"], weight_decay=g["weight_decay"], step=self.step_count, max_norm=self.max_norm)
		self.model.mark_shadow_fresh()
		return self.grad_norm

	def _layout(self) -> list:
		"""(name, offset, shape, storage elements) of every tensor in the flat moment buffers: the decoder's flat parameter layout.  It is NOT a function of the
		constructor arguments alone across versions of this package -- round 5 began to store the tied logits matrix with ceil(V / 64) x 64 rows and appended bias / scale /
		hidden-norm rows -- so the moments travel with the table that says where each tensor sits."""
		offs = self.model._offsets
		names = list(offs)
		table = []
		for i, n in enumerate(names):
			o, shape = offs[n]
			end = offs[names[i + 1]][0] if i + 1 < len(names) else self.model._n_flat
			table.append((n, int(o), tuple(int(d) for d in shape), int(end - o)))
		return table

	def state_dict(self) -> dict[str, Any]:
		return dict(step=self.step_count, exp_avg=self.exp_avg.clone(), exp_avg_sq=self.exp_avg_sq.clone(), param_groups=[dict(g) for g in self.param_groups], max_norm=self.max_norm,
		            layout=self._layout())

	def load_state_dict(self, state: dict[str, Any]):
		"""Moments are placed BY NAME through the `layout` table the state carries (any flat layout of the same tensors loads: another storage row count of the logits
		matrix, another order); a state without the table (written before round 6) must have this model's flat size exactly, and says so when it has not."""
		mine = self._layout()
		theirs = state.get("layout")
		src_m, src_v = state["exp_avg"], state["exp_avg_sq"]
		if theirs is None:
			if src_m.numel() != self.exp_avg.numel() or src_v.numel() != self.exp_avg_sq.numel():
				raise ValueError(f"FusedAdamW state without a layout table holds {src_m.numel()} moment elements, this model's flat layout has {self.exp_avg.numel()}: it was written "
				                 "for another parameter layout (e.g. before the logits matrix got its storage rows) and cannot be placed safely")
			self.exp_avg.copy_(src_m)
			self.exp_avg_sq.copy_(src_v)
		elif [(t[0], int(t[1]), tuple(t[2]), int(t[3])) for t in theirs] == mine:
			self.exp_avg.copy_(src_m)
			self.exp_avg_sq.copy_(src_v)
		else:
			there = {t[0]: (t[0], int(t[1]), tuple(t[2]), int(t[3])) for t in theirs}
			missing = [n for n, *_ in mine if n not in there]
			extra = [n for n in there if n not in {m[0] for m in mine}]
			if missing or extra:
				raise ValueError(f"FusedAdamW state is for another model: missing {missing[:4]}{'...' if len(missing) > 4 else ''}, unexpected {extra[:4]}{'...' if len(extra) > 4 else ''}")
			self.exp_avg.zero_()
			self.exp_avg_sq.zero_()
			for name, o, shape, _ in mine:
				_, so, sshape, _ = there[name]
				if tuple(sshape) != tuple(shape):
					raise ValueError(f"FusedAdamW state: {name} has shape {tuple(sshape)} in the checkpoint, {tuple(shape)} in this model")
				n = math.prod(shape)
				self.exp_avg[o:o + n].copy_(src_m[so:so + n])
				self.exp_avg_sq[o:o + n].copy_(src_v[so:so + n])
		self.step_count = state["step"]
		self.param_groups = [dict(g) for g in state["param_groups"]]
		self.max_norm = state.get("max_norm", self.max_norm)

	def load_reference_state_dict(self, state: dict[str, Any], param_order: Sequence[str]):
		"""The `optimizer_state_dict` of a `.train` file the REFERENCE wrote (`torch.optim.AdamW.state_dict()`, train.py:1465) into the flat moments.

		torch numbers the parameters group by group; the reference's groups are (train.py:1103-1119) the < 2-D tensors of `model.parameters()` with weight decay 0, then the
		>= 2-D ones with the configured decay -- or one group of everything with `weight_decay_1d`.  `param_order` is that `model.parameters()` order: the key order of the
		checkpoint's own `model_state_dict` (a module's state dict lists parameters in registration order), buffers left out.  Every shape is checked against the slot it lands in."""
		offsets = self.model._offsets
		groups = state["param_groups"]
		shapes = {n: tuple(offsets[n][1]) for n in param_order}
		if len(groups) == 1:
			order = list(param_order)
		elif len(groups) == 2:
			order = [n for n in param_order if len(shapes[n]) < 2] + [n for n in param_order if len(shapes[n]) >= 2]
			if len(groups[0]["params"]) != sum(len(shapes[n]) < 2 for n in param_order):
				raise ValueError("Reference optimizer state: the first parameter group does not hold the < 2-D tensors (train.py:1103-1119 layout expected)")
		else:
			raise ValueError(f"Reference optimizer state has {len(groups)} parameter groups; the reference writes one or two")
		ids = [i for g in groups for i in g["params"]]
		if len(ids) != len(order):
			raise ValueError(f"Reference optimizer state covers {len(ids)} tensors, the model has {len(order)}")
		steps = set()
		self.exp_avg.zero_()
		self.exp_avg_sq.zero_()
		for idx, name in zip(ids, order):
			st = state["state"].get(idx)
			if st is None:  # a tensor that never received a gradient has no entry: moments stay zero
				continue
			o, shape = offsets[name]
			n = math.prod(shape)
			for key, dst in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
				if tuple(st[key].shape) != tuple(shape):
					raise ValueError(f"Reference optimizer state: {key} of parameter {idx} has shape {tuple(st[key].shape)}, {name} is {tuple(shape)}")
				dst[o:o + n].copy_(st[key].reshape(-1))
			steps.add(int(st["step"]))
		if len(steps) > 1:
			raise ValueError(f"Reference optimizer state: tensors disagree on the step count ({sorted(steps)}); the fused update keeps one")
		self.step_count = steps.pop() if steps else 0
		decay = groups[-1]
		g0 = self.param_groups[0]
		g0.update(lr=float(decay["lr"]), initial_lr=float(decay.get("initial_lr", decay["lr"])), betas=tuple(decay["betas"]), eps=float(decay["eps"]),
		          weight_decay=float(decay["weight_decay"]))
		self.weight_decay_1d = len(groups) == 1 and float(decay["weight_decay"]) != 0.0


class ChunkSchedule:
	"""LinearLR warm-up (start factor 1/(w+1), w chunks) chained with CosineAnnealingLR(T_max, eta_min), both stepped once per chunk,
	warm-up first (reference train.py:1138-1158, :1339-1342).  Restates the chainable recursions of the two torch schedulers, so the
	learning rate sequence is the reference's, including the (non-product) interaction of the two when eta_min != 0."""

	def __init__(self, optimizer: FusedAdamW, base_lr: float, warmup_chunks: int, scheduler: str, t_max: int, final_lr: float):
		self.opt, self.base_lr, self.warmup, self.kind, self.t_max, self.final_lr = optimizer, base_lr, warmup_chunks, scheduler.lower(), max(t_max, 1), final_lr
		if self.kind not in ("const", "cosine"):
			raise ValueError(f"Unsupported learning rate scheduler: {scheduler}")
		self.chunks_done = 0
		self.current = base_lr * (1.0 / (warmup_chunks + 1) if warmup_chunks >= 1 else 1.0)
		self._apply()

	def _apply(self):
		self.opt.param_groups[0]["lr"] = self.current

	def step(self):
		self.chunks_done += 1
		n, lr = self.chunks_done, self.current
		if self.warmup >= 1 and n <= self.warmup:
			start = 1.0 / (self.warmup + 1)
			lr *= 1.0 + (1.0 - start) / (self.warmup * start + (n - 1) * (1.0 - start))
		if self.kind == "cosine":
			T, eta = self.t_max, self.final_lr
			if (n - 1 - T) % (2 * T) == 0:
				lr += (self.base_lr - eta) * (1 - math.cos(math.pi / T)) / 2
			else:
				lr = (1 + math.cos(math.pi * n / T)) / (1 + math.cos(math.pi * (n - 1) / T)) * (lr - eta) + eta
		self.current = lr
		self._apply()

	def state_dict(self):
		return dict(chunks_done=self.chunks_done, current=self.current, base_lr=self.base_lr, warmup=self.warmup, kind=self.kind, t_max=self.t_max, final_lr=self.final_lr)

	def load_state_dict(self, s):
		self.chunks_done, self.current, self.base_lr, self.warmup = s["chunks_done"], s["current"], s["base_lr"], s["warmup"]
		self.kind, self.t_max, self.final_lr = s["kind"], s["t_max"], s["final_lr"]
		self._apply()

	def load_reference_state_dicts(self, warmup_sd: Optional[dict], scheduler_sd: Optional[dict]):
		"""`scheduler_warmup_state_dict` (torch LinearLR) / `scheduler_state_dict` (torch CosineAnnealingLR) of a `.train` file the reference wrote (train.py:1466-1467): both
		are stepped once per chunk (:1339-1342), so either's `last_epoch` is the number of chunks done; `_last_lr` of the one stepped last is the rate in force."""
		last = scheduler_sd or warmup_sd
		if last is None:
			return
		self.chunks_done = int(last["last_epoch"])
		self.current = float(last["_last_lr"][0])
		self.base_lr = float(last["base_lrs"][0])
		if warmup_sd is not None:
			self.warmup = int(warmup_sd["total_iters"])
		if scheduler_sd is not None:
			self.kind, self.t_max, self.final_lr = "cosine", int(scheduler_sd["T_max"]), float(scheduler_sd["eta_min"])
		self._apply()


# ------------------------------------------------------------------------------------------------------------------------------
# data parallel
# ------------------------------------------------------------------------------------------------------------------------------

class DataParallel:
	"""One process per GPU; gradients summed once per optimizer step over RCCL/xGMI (backend 'nccl' on ROCm) or gloo on CPU tests."""

	def __init__(self, buckets: int = 2, persistent_cus: Optional[int] = None):
		"""persistent_cus: workgroups the persistent 256-wide GEMM grids AND the weight-gradient launches (ABI 9: novic_wgrad*_bf16 take the budget too -- their tiles x parts
		grids are the largest of the backward pass) may have WHILE early all-reduces are in flight (`ops.cu_budget`, a per-call argument of the C ABI).  Those grids otherwise own every CU with all of its LDS and registers, and an RCCL kernel launched beside them waits for a whole grid to
		end -- or holds CUs the grid's last workgroups queue for; 8-16 workgroups short leaves the collective CUs of its own.  None: $NOVIC_DP_PERSISTENT_CUS, else the
		library default (256 = no reservation, or $NOVIC_PERSISTENT_CUS).  UNMEASURED: this pool has no multi-GPU node; `bench.py --persistent-cus N` exists so that the
		first 8-GPU session can A/B it."""
		import torch.distributed as dist
		self.dist = dist
		self.enabled = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
		self.world = dist.get_world_size() if self.enabled else 1
		self.rank = dist.get_rank() if self.enabled else 0
		self.buckets = max(1, buckets)
		if persistent_cus is None and os.environ.get("NOVIC_DP_PERSISTENT_CUS"):
			persistent_cus = int(os.environ["NOVIC_DP_PERSISTENT_CUS"])
		if persistent_cus is not None and not 8 <= int(persistent_cus) <= 256:
			raise ValueError("DataParallel(persistent_cus): 8..256 workgroups")
		self.persistent_cus = int(persistent_cus) if persistent_cus is not None else None
		# First-contact instrumentation for a multi-GPU node (bench.py switches it on for a pass of its own, never inside the timed region): per optimizer step a HIP event
		# pair around the END of the exchange -- recorded on the compute stream behind the last backward kernel and behind the last collective's wait, i.e. the stream time
		# between the backward pass and the optimizer launch that the early per-layer reductions did NOT hide -- and the bytes that went out early / in the tail.
		self.instrument = False
		self.exposed_events: list = []
		self.bytes_early = self.bytes_tail = 0

	def decorrelate(self, model, embed_noise=None):
		"""Give this rank its own dropout-mask and noise streams (idempotent): the seeds are the constructor's, so without this every rank would draw the
		SAME masks and the same noise for its different samples.  Rank 0 keeps the single-process streams."""
		if getattr(model, "_dp_rank_folded", None) != self.rank:
			base = getattr(model, "_dropout_seed_base", model.dropout_seed)
			model._dropout_seed_base = base
			model.dropout_seed = (base + 0xD1B54A32D192ED03 * self.rank) & 0xFFFFFFFFFFFFFFFF
			model._dp_rank_folded = self.rank
		if embed_noise is not None and getattr(embed_noise, "_dp_rank_folded", None) != self.rank:
			base = getattr(embed_noise, "_seed_base", embed_noise.seed)
			embed_noise._seed_base = base
			embed_noise.seed = (base + 0x9E3779B97F4A7C15 * self.rank) & 0xFFFFFFFFFFFFFFFF
			embed_noise._dp_rank_folded = self.rank

	def begin_step(self):
		"""Forget the early reductions of the previous optimizer step."""
		self._early: list = []

	def reduce_range_early(self, flat_grad: torch.Tensor, start: int, end: int):
		"""Called from inside the backward pass as soon as flat_grad[start:end] is final (a layer's weight gradients): the SUM all-reduce of that
		slice is enqueued behind the kernels that produce it and runs on RCCL's stream UNDER the rest of the backward pass."""
		if not self.enabled or end <= start:
			return
		work = self.dist.all_reduce(flat_grad[start:end], op=self.dist.ReduceOp.SUM, async_op=True)
		self._early.append((start, end, work))
		if self.instrument:
			self.bytes_early += (end - start) * flat_grad.element_size()
		if self.persistent_cus is not None and not getattr(self, "_budget_on", False):
			# from the first collective in flight to all_reduce_grads(): the GEMM grids of the rest of the backward pass leave CUs free for RCCL's kernels
			self._budget_prev = ops.set_cu_budget(self.persistent_cus)
			self._budget_on = True

	def _restore_budget(self):
		if getattr(self, "_budget_on", False):
			ops.set_cu_budget(self._budget_prev)
			self._budget_on = False

	def all_reduce_grads(self, flat_grad: torch.Tensor):
		"""SUM all-reduce of the flat gradient buffer (the loss was pre-scaled by 1/world): whatever reduce_range_early() has not covered yet goes out
		as up to `buckets` async collectives per gap, then every outstanding collective is waited for.  46.7 MB per step for the default 11.7 M-
		parameter decoder, of which 28.3 MB (the six layers) are already in flight under the backward pass."""
		self._restore_budget()
		if not self.enabled:
			return
		early = sorted(getattr(self, "_early", []), key=lambda t: t[0])
		works = [w for _, _, w in early]
		gaps, pos = [], 0
		for s0, e0, _ in early:
			if s0 > pos:
				gaps.append((pos, s0))
			pos = max(pos, e0)
		if pos < flat_grad.numel():
			gaps.append((pos, flat_grad.numel()))
		ev0 = ev1 = None
		if self.instrument and flat_grad.is_cuda:
			ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
			ev0.record()  # behind the last kernel of the backward pass on the compute stream
			self.bytes_tail += sum(e0 - s0 for s0, e0 in gaps) * flat_grad.element_size()
		for s0, e0 in gaps:
			n = e0 - s0
			k = self.buckets if n >= (1 << 20) else 1
			edges = [s0 + n * i // k for i in range(k + 1)]
			works += [self.dist.all_reduce(flat_grad[edges[i]:edges[i + 1]], op=self.dist.ReduceOp.SUM, async_op=True) for i in range(k) if edges[i + 1] > edges[i]]
		for w in works:
			w.wait()
		if ev1 is not None:
			ev1.record()  # the compute stream has been told to wait for every collective: what follows is the optimizer launch
			self.exposed_events.append((ev0, ev1))
		self._early = []

	def exposed_ms_per_step(self) -> Optional[float]:
		"""Mean stream time between the end of the backward pass and the end of the exchange over the instrumented steps so far (synchronises); None without any."""
		if not self.exposed_events:
			return None
		torch.cuda.synchronize()
		ms = [a.elapsed_time(b) for a, b in self.exposed_events]
		return sum(ms) / len(ms)

	def reset_instrument(self, on: bool):
		self.instrument, self.exposed_events, self.bytes_early, self.bytes_tail = bool(on), [], 0, 0

	def broadcast_parameters(self, flat: torch.Tensor):
		if self.enabled:
			self.dist.broadcast(flat, src=0)

	def all_reduce_stats(self, stats: torch.Tensor):
		if self.enabled:
			self.dist.all_reduce(stats, op=self.dist.ReduceOp.SUM)


# ------------------------------------------------------------------------------------------------------------------------------
# one optimizer step
# ------------------------------------------------------------------------------------------------------------------------------

def train_step(model: embedding_decoder.PrefixedIterDecoder, optimizer: FusedAdamW, micro_batches, *, embed_noise: Optional[embedding_noise.EmbeddingNoise] = None,
               dp: Optional[DataParallel] = None, merged: bool = True):
	"""noise -> forward -> backward over this rank's micro-batches -> gradient all-reduce -> clip -> AdamW.

	micro_batches: list of (embed B x F, target, mask, weight) device tensors of one optimizer step (all the same shapes when merged).
	Returns (stats 4 x n_micro device tensor [basis, loss_sum, correct, tokens], grad-norm device tensor).
	"""
	world = dp.world if dp is not None else 1
	accum = len(micro_batches)
	scale = 1.0 / (accum * world)
	optimizer.zero_grad()
	if dp is not None:
		dp.begin_step()
	single_pass = (merged and accum > 1 and _mergeable(model, micro_batches)) or accum == 1
	# one backward pass per step: a layer's weight gradients are final when its backward is enqueued -> reduce them under the rest of the pass
	model.grad_ready_hook = (lambda start, end: dp.reduce_range_early(model.flat_grad(), start, end)) if (dp is not None and dp.enabled and single_pass) else None
	try:
		if merged and accum > 1 and _mergeable(model, micro_batches):
			whole = _group_of(micro_batches)
			if whole is not None:  # the loader assembled the step's micro-batches in one set of buffers (embedding_cache.GroupSlice): nothing to concatenate
				embed, target, mask, weight = whole
			else:
				embed = torch.cat([mb[0] for mb in micro_batches], dim=0)
				target = torch.cat([mb[1] for mb in micro_batches], dim=0)
				mask = None if micro_batches[0][2] is None else torch.cat([mb[2] for mb in micro_batches], dim=0)
				weight = None if micro_batches[0][3] is None else torch.cat([mb[3] for mb in micro_batches], dim=0)
			if embed_noise is not None:
				embed = embed_noise(embed)
			stats = model.forward_backward(embed, target, mask, weight, group_rows=micro_batches[0][0].shape[0], loss_scale=scale)
		else:
			parts = []
			for embed, target, mask, weight in micro_batches:
				if embed_noise is not None:
					embed = embed_noise(embed)
				parts.append(model.forward_backward(embed, target, mask, weight, loss_scale=scale).clone())
			stats = torch.cat(parts, dim=1)
	finally:
		if dp is not None:
			dp._restore_budget()  # (an exception inside the pass must not leave this thread's GEMM grids on the reduced budget)
	model.grad_ready_hook = None
	if dp is not None:
		dp.all_reduce_grads(model.flat_grad())
	return stats, optimizer.step()


def _group_of(micro_batches):
	"""The buffers behind the micro-batches when they are ALL the slices of one loader group, in order (embedding_cache.GroupSlice); None otherwise."""
	first = micro_batches[0]
	full = getattr(first, "full", None)
	if full is None or getattr(first, "size", 0) != len(micro_batches):
		return None
	for i, mb in enumerate(micro_batches):
		if getattr(mb, "full", None) is not full or mb.pos != i:
			return None
	return full


def _mergeable(model, micro_batches) -> bool:
	"""Micro-batches can be stacked along the sample dimension when they share shapes (and the sample dimension comes first)."""
	first = micro_batches[0]
	if first[1].ndim == 3 and model.data_config.multi_target and model.data_config.multi_first:
		return False
	for mb in micro_batches[1:]:
		if mb[0].shape != first[0].shape or mb[1].shape != first[1].shape or (mb[2] is None) != (first[2] is None) or (mb[3] is None) != (first[3] is None):
			return False
	return True


# ------------------------------------------------------------------------------------------------------------------------------
# training loop (reference train.py:927-974, :1193-1429)
# ------------------------------------------------------------------------------------------------------------------------------

@dataclasses.dataclass(frozen=True)
class TrainLoopConfig:
	run_dir: str
	wandb: bool
	save_every_min: int
	save_every_max: int
	save_top1_min: float
	save_top1_delta: float
	gradient_clip: float
	last_dropout_chunks: int
	last_dropout_factor: float
	device_is_cpu: bool
	epoch_batches: int
	chunk_batches: int
	chunk_samples: int
	max_chunks: int
	ewa_factor: float
	ewa_factor_inv: float


@dataclasses.dataclass(eq=False)
class TrainLoopState:
	epoch_id: int = 1
	chunk_id: int = 1
	batch_id: int = 1
	sample_id: int = 1
	epoch_started: bool = False
	chunk_started: bool = False
	epoch_batches_left: int = -1
	start_time: Optional[float] = None
	epoch_start_time: Optional[float] = None
	chunk_start_time: Optional[float] = None
	num_grad_norms: int = 0
	grad_norms: Optional[torch.Tensor] = None
	ewa_train_loss_sum: float = 0.0
	ewa_train_loss_basis: float = 0.0
	ewa_train_loss: Optional[float] = None
	ewa_train_correct: float = 0.0
	ewa_train_tokens: float = 0.0
	ewa_train_top1: float = 0.0
	ewa_train_top1_max: float = 0.0
	ewa_train_top1_last: float = 0.0
	allow_save_delta: bool = False
	saved_num: int = 0
	saved_chunk_id: int = 0
	saved_ewa_train_loss: float = math.inf
	saved_ewa_train_top1: float = 0.0
	saved_ewa_train_top1_max: float = 0.0


def make_train_loop_config(*, run_dir: str, batch_size: int, epoch_batches: int, num_valid_targets: int, accum_size: int, chunk_scale: float = 50, max_chunks: int = 0,
                           max_epochs: int = 18, save_every_min: int = 12, save_every_max: int = 48, save_top1_min: float = 95.0, save_top1_delta: float = 0.5,
                           gradient_clip: float = 1.0, loss_ewa_halflife: float = 4, last_dropout_chunks: int = 0, last_dropout_factor: float = 0.0, use_wandb: bool = False,
                           device_is_cpu: bool = False) -> TrainLoopConfig:
	"""Chunk / epoch arithmetic of action_train (reference train.py:987-1000, :1036-1053)."""
	chunk_batches = max(math.ceil(num_valid_targets * chunk_scale / batch_size), accum_size, 1)
	limit = sys.maxsize - 1
	if limit > max_chunks >= 1:
		limit = max_chunks
	if max_epochs >= 1 and limit > (by_epochs := (max_epochs * epoch_batches) // chunk_batches):
		limit = by_epochs
	ewa = 0.5 ** (1 / (loss_ewa_halflife * chunk_batches))
	return TrainLoopConfig(run_dir=run_dir, wandb=use_wandb, save_every_min=save_every_min, save_every_max=save_every_max, save_top1_min=save_top1_min / 100,
	                       save_top1_delta=save_top1_delta / 100, gradient_clip=gradient_clip, last_dropout_chunks=last_dropout_chunks, last_dropout_factor=last_dropout_factor,
	                       device_is_cpu=device_is_cpu, epoch_batches=epoch_batches, chunk_batches=chunk_batches, chunk_samples=chunk_batches * batch_size, max_chunks=limit,
	                       ewa_factor=ewa, ewa_factor_inv=1 - ewa)


def rescale_dropout(model: embedding_decoder.PrefixedIterDecoder, factor: float):
	assert factor >= 0
	model.input_dropout *= factor
	model.layer_dropout *= factor


def training_loop(cfg_flat: dict[str, Any], C: TrainLoopConfig, S: TrainLoopState, model: embedding_decoder.PrefixedIterDecoder, target_nouns: tuple[str, ...],
                  num_invalid_target_nouns: int, mean_shift: Optional[torch.Tensor], embed_noise: Optional[embedding_noise.EmbeddingNoise],
                  grad_accum: embedding_dataset.GradAccum, optimizer: FusedAdamW, schedule: Optional[ChunkSchedule], device: torch.device, dp: Optional[DataParallel] = None,
                  log: Callable[[str], None] = print, on_chunk: Optional[Callable[[dict], None]] = None, rng_state: Optional[Callable[[], dict]] = None):
	"""Trains until C.max_chunks chunks are done.  Same state machine as the reference loop; metrics are replayed per chunk."""
	S.start_time = time.perf_counter()
	stop = S.chunk_id >= C.max_chunks + 1
	if C.last_dropout_chunks >= 1 and S.chunk_id > C.max_chunks - C.last_dropout_chunks:
		rescale_dropout(model, C.last_dropout_factor)
	assert C.epoch_batches >= 1
	if S.epoch_batches_left < 0:
		S.epoch_batches_left = C.epoch_batches
	elif S.epoch_batches_left == 0:
		S.epoch_batches_left = C.epoch_batches
		S.epoch_id += 1
		S.epoch_started = False
	if mean_shift is not None:
		if embed_noise is None:
			embed_noise = embedding_noise.MeanShiftOnly(model.embed_dim, mean_shift.to(device))
		else:
			embed_noise.mean_shift = mean_shift.reshape(-1).contiguous().to(device)
	rank0 = dp is None or dp.rank == 0
	pending_stats: list[torch.Tensor] = []   # device tensors, one 4 x accum block per optimizer step of the current chunk
	pending_norms: list[torch.Tensor] = []
	pending_sizes: list[int] = []

	def finish_chunk():
		nonlocal stop
		stats = torch.cat(pending_stats, dim=1)
		if dp is not None:
			dp.all_reduce_stats(stats)
		stats = stats.cpu()  # the chunk's single host synchronisation
		# (taken BEHIND that synchronisation: the chunk's optimizer steps are enqueued asynchronously, and until round 4 the clock was read before the device had run them --
		# the logged noun/s was the host's enqueue rate, three times the real one; the reference's per-batch .item() calls keep its clock honest, train.py:1288-1305)
		elapsed = time.perf_counter() - S.chunk_start_time
		norms = torch.cat(pending_norms).cpu() if pending_norms else torch.zeros(0)
		for i in range(stats.shape[1]):  # replay of the per-batch EWA recursion (reference :1288-1305)
			basis, loss_sum, correct, tokens = (float(stats[k, i]) for k in range(4))
			S.ewa_train_loss_sum = S.ewa_train_loss_sum * C.ewa_factor + loss_sum
			S.ewa_train_loss_basis = S.ewa_train_loss_basis * C.ewa_factor + basis
			S.ewa_train_loss = S.ewa_train_loss_sum / S.ewa_train_loss_basis
			S.ewa_train_correct = S.ewa_train_correct * C.ewa_factor + correct
			S.ewa_train_tokens = S.ewa_train_tokens * C.ewa_factor + tokens
			S.ewa_train_top1 = S.ewa_train_correct / S.ewa_train_tokens
			S.ewa_train_top1_max = max(S.ewa_train_top1_max, S.ewa_train_top1)
		pending_stats.clear(); pending_norms.clear(); pending_sizes.clear()
		world = dp.world if dp is not None else 1
		info = dict(chunk=S.chunk_id, lr=optimizer.lr, loss=S.ewa_train_loss, top1=S.ewa_train_top1, chunk_time=elapsed, samples_per_s=C.chunk_samples * world / elapsed)
		if norms.numel():
			info.update(grad_norm_min=float(norms.min()), grad_norm_mean=float(norms.mean()), grad_norm_max=float(norms.max()))
		if rank0:
			log(f"Trained chunk {S.chunk_id} in {elapsed:.1f}s at {info['samples_per_s']:.0f}noun/s: lr={optimizer.lr:.2e}, loss={S.ewa_train_loss:.2e}, top1={S.ewa_train_top1:.3%}")
		if schedule is not None:
			schedule.step()
		S.chunk_id += 1
		S.chunk_started = False
		if S.chunk_id >= C.max_chunks + 1:
			stop = True
		save_chunk_id = S.chunk_id - 1
		since = save_chunk_id - S.saved_chunk_id
		if S.ewa_train_top1 >= C.save_top1_min and S.ewa_train_top1 - S.ewa_train_top1_last <= C.save_top1_delta:
			S.allow_save_delta = True
		S.ewa_train_top1_last = S.ewa_train_top1
		if stop or since >= C.save_every_max or (since >= C.save_every_min and S.ewa_train_top1 >= C.save_top1_min and S.allow_save_delta and S.ewa_train_top1 >= S.saved_ewa_train_top1_max):
			S.saved_num += 1
			S.saved_chunk_id = save_chunk_id
			S.saved_ewa_train_loss = S.ewa_train_loss
			S.saved_ewa_train_top1 = S.ewa_train_top1
			S.saved_ewa_train_top1_max = max(S.saved_ewa_train_top1_max, S.ewa_train_top1)
			if rank0 and C.run_dir:
				extra = dict(noise_calls=embed_noise.calls) if embed_noise is not None else {}
				if rng_state is not None:
					extra.update(rng_state())
				path = save_train_checkpoint(cfg_flat=cfg_flat, model=model, C=C, S=S, target_nouns=target_nouns, num_invalid_target_nouns=num_invalid_target_nouns,
				                             optimizer=optimizer, schedule=schedule, extra_rng_state=extra)
				log(f"Saved checkpoint: {path}")
		if C.last_dropout_chunks >= 1 and S.chunk_id == C.max_chunks - C.last_dropout_chunks + 1:
			rescale_dropout(model, C.last_dropout_factor)
		if on_chunk is not None:
			on_chunk(info)

	while not stop:
		S.epoch_start_time = time.perf_counter()
		model.train()
		step_batches = []
		for batch in grad_accum.loader():
			S.epoch_started = S.chunk_started = True
			if (S.batch_id - 1) % C.chunk_batches == 0:
				S.chunk_start_time = time.perf_counter()
			step_batches.append(batch)
			_, do_step = grad_accum.loss_scale(batch[0].shape[0])
			if do_step:
				stats, norm = train_step(model, optimizer, step_batches, embed_noise=embed_noise, dp=dp)
				pending_stats.append(stats.clone())
				pending_norms.append(norm.clone())
				step_batches = []
			S.sample_id += batch[0].shape[0]
			S.batch_id += 1
			S.epoch_batches_left -= 1
			if (S.batch_id - 2) % C.chunk_batches == C.chunk_batches - 1:
				if step_batches:  # chunk boundary inside an accumulation window cannot happen: chunk_batches >= accum_size and both count loader batches
					pass
				finish_chunk()
				if stop:
					break
			if S.epoch_batches_left == 0:
				break
		if S.epoch_batches_left == 0:
			S.epoch_batches_left = C.epoch_batches
			S.epoch_id += 1
			S.epoch_started = False
	if not S.epoch_started:
		S.epoch_id -= 1
	if not S.chunk_started:
		S.chunk_id -= 1
	S.batch_id -= 1
	S.sample_id -= 1
	if rank0:
		log(f"Trained for {S.chunk_id} chunks (up to {S.epoch_id} epochs) in {time.perf_counter() - S.start_time:.1f}s = {S.batch_id} batches = {S.sample_id} samples")


def save_train_checkpoint(cfg_flat: dict[str, Any], model: embedding_decoder.EmbeddingDecoder, C: Optional[TrainLoopConfig], S: Optional[TrainLoopState],
                          target_nouns: tuple[str, ...], num_invalid_target_nouns: int, optimizer: Optional[FusedAdamW], schedule: Optional[ChunkSchedule], *,
                          model_only: bool = False, run_dir: Optional[str] = None, chunk_id: Optional[int] = None, extra_rng_state: Optional[dict] = None) -> str:
	"""Same dict layout as reference train.py:1450-1473 (cfg_flat, target_config, data_config, model_state_dict, target_nouns,
	num_invalid_target_nouns [+ .train extras]); optimizer state is this build's flat exp_avg / exp_avg_sq."""
	ckpt = dict(
		cfg_flat=cfg_flat,
		target_config=dataclasses.asdict(model.target_config),
		data_config=dataclasses.asdict(model.data_config),
		model_state_dict={k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
		target_nouns=target_nouns,
		num_invalid_target_nouns=num_invalid_target_nouns,
	)
	path = os.path.join(C.run_dir if run_dir is None else run_dir, f"ovod_chunk{S.saved_chunk_id if chunk_id is None else chunk_id:04d}_{datetime.datetime.now().strftime('%Y%m%d_%H%M%S')}")
	if model_only:
		path += ".model"
	else:
		path += ".train"
		state = dataclasses.asdict(S)
		ckpt.update(
			train_loop_config=dataclasses.asdict(C),
			train_loop_state=state,
			optimizer_type="novic_amd.train.FusedAdamW",
			optimizer_state_dict={k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in optimizer.state_dict().items()},
			scheduler_warmup_state_dict=None,
			scheduler_state_dict=schedule and schedule.state_dict(),
			amp_scaler_enabled=False,
			amp_scaler_state_dict={},
			# streams the reference leaves to torch's global generator: dropout-mask and noise call counters (and, when the loop hands it over, the loader's
			# shuffle generator) -- a resumed run continues them instead of replaying the first chunk's masks and batch order
			novic_rng_state=dict(dropout_calls=getattr(model, "_dropout_calls", 0), **(extra_rng_state or {})),
		)
	os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
	torch.save(ckpt, path)
	return path


# ------------------------------------------------------------------------------------------------------------------------------
# action_train (reference train.py:977-1190 and its helpers :3587, :3631, :3714, :3741, :3873-3957, :3969): the cfg-driven assembly
# embedder -> cache dataset -> target / data config -> loader -> GradAccum -> mean shift -> noise -> model -> optimizer -> schedule -> resume -> loop
# ------------------------------------------------------------------------------------------------------------------------------

# config/train.yaml key names and defaults of everything the train action reads (the flag contract; hydra itself is not needed: `cfg` is any object with these
# attributes -- an omegaconf.DictConfig of the reference's yaml, or utils.AttrDict(default_train_config(), **overrides))
_TRAIN_DEFAULTS = dict(
	action="train", device="cuda", determ=False, determ_seed=1, dry_run=False, wandb=False,
	embedder_spec="", embedder_amp=True, embedder_amp_bf16=False, embedder_compile=False, embedder_optimum=False, batch_size_token=2048, batch_size_embed=512, batch_size_image=256,
	embedding_dataset="", embedding_cache_dir="", strict_embedder=True, batch_size=512, dataset_workers=8,
	load_model="", model="PrefixedIterDecoder", with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True,
	use_weights=False, multi_target=False, multi_first=False, fixed_multi_length=False, amp=False, amp_bf16=True,
	vocab_quant=False, num_end_loss=1, label_smoothing=0.0, hidden_dim=512, feedfwd_scale="1/4", mlp_hidden_layer="none", mlp_hidden_bias=False, mlp_hidden_norm=False,
	mlp_hidden_activation="gelu", input_dropout=0.1, num_layers=6, num_heads=8, layer_dropout=0.1, layer_activation="gelu", layer_norm_first=True, layer_bias=False,
	logits_bias=False, init_bias_zero=True, init_mlp_mode="balanced", init_mlp_unit_norm=False, init_tfrm_mode="balanced", init_tfrm_unit_norm=False,
	init_tfrm_unit_postnorm=True, init_tfrm_proj_layers=True, init_zero_norm=False, init_rezero_mode="none", mlp_seq_len=4, weight_tying=True, strictly_causal=False,
	enable_nested=False,
	load_train_state=True, load_lr_state=True, chunk_scale=50, save_every_min=12, save_every_max=48, save_top1_min=95.0, save_top1_delta=0.5, max_epochs=18, max_chunks=0,
	accum_factor=8, optimizer="AdamW", init_lr=1.5e-3, final_lr=0.0, lr_scheduler="cosine", lr_warmup=0, beta1=0.9, beta2=0.95, weight_decay=0.1, weight_decay_1d=False,
	nesterov=True, compile=False, gradient_clip=1.0, loss_ewa_halflife=4, last_dropout_chunks=0, last_dropout_factor=0.0, mean_shift=False,
	mean_shift_path="$SOURCE/data/modality_gap_$EMBEDDER.json", noise_scheme="", noise_vec_norm=0.0, noise_angle_min=0.0, noise_angle_max=0.0, noise_angle_std=0.0,
	noise_mix_ratio=0.0,
)
IGNORE_CFG_DIFFS = {"action", "device", "wandb", "load_model", "load_train_state", "load_lr_state", "max_chunks", "max_epochs", "dry_run", "embedding_dataset", "embedding_cache_dir"}


def default_train_config(**overrides) -> utils.AttrDict:
	"""The train action's slice of config/train.yaml at its defaults, with overrides (CLI `key=value` in the reference)."""
	unknown = set(overrides) - set(_TRAIN_DEFAULTS)
	if unknown:
		raise ValueError(f"Unknown train configuration keys: {sorted(unknown)}")
	return utils.AttrDict(dict(_TRAIN_DEFAULTS, **overrides))


def _flat_cfg(cfg) -> dict[str, Any]:
	"""utils_config.flatten_config: what the checkpoint stores as cfg_flat (and infer.NOVICModel rebuilds its cfg from)."""
	if isinstance(cfg, dict):
		return utils.flatten_dict(dict(cfg))
	try:
		import omegaconf
		return utils.flatten_dict(omegaconf.OmegaConf.to_container(cfg, resolve=True))
	except ImportError:
		return utils.flatten_dict({k: getattr(cfg, k) for k in _TRAIN_DEFAULTS if hasattr(cfg, k)})


def safe_embedder_spec(spec: str) -> str:
	return "".join(ch if ch.isalnum() or ch in "-_." else "_" for ch in spec)


def resolve_source_path(path: str, source_dir: Optional[str] = None) -> str:
	"""`$SOURCE` = the checkout the data files live in (reference train.py:62, :4271); here the NOVIC_SOURCE environment variable or the working directory."""
	return os.path.expanduser(path.replace("$SOURCE", source_dir or os.environ.get("NOVIC_SOURCE", os.getcwd())))


def check_loaded_config(name: str, using: dict[str, Any], loaded: dict[str, Any], ignore: Iterable[str] = (), log: Callable[[str], None] = print) -> bool:
	"""Warns about every difference between a configuration in use and the one a checkpoint was written with (reference train.py:3912-3957); True = identical."""
	ignore = set(ignore)
	issues = []
	for key in sorted((set(using) | set(loaded)) - ignore):
		if key not in loaded:
			issues.append(f"loaded {name} has no '{key}'")
		elif key not in using:
			issues.append(f"loaded {name} has unused '{key}'")
		else:
			a, b = using[key], loaded[key]
			if isinstance(a, torch.Tensor) or isinstance(b, torch.Tensor):
				same = isinstance(a, torch.Tensor) and isinstance(b, torch.Tensor) and a.dtype == b.dtype and a.shape == b.shape and torch.equal(a.cpu(), b.cpu())
			else:
				same = type(a) is type(b) and a == b
			if not same:
				issues.append(f"{name} '{key}': loaded {b!r} vs using {a!r}")
	for msg in issues:
		log(f"WARNING: {msg}")
	if issues:
		log(f"WARNING: mismatches between the loaded and the in-use {name} => verify this is okay")
	return not issues


def load_embedder(cfg, device: torch.device, load_model: bool = False) -> embedders.Embedder:
	return embedders.Embedder.create(spec=cfg.embedder_spec, amp=cfg.embedder_amp, amp_bf16=cfg.embedder_amp_bf16, tokenizer_batch_size=cfg.batch_size_token,
	                                 inference_batch_size=cfg.batch_size_embed, image_batch_size=cfg.batch_size_image, load_model=load_model,
	                                 compile_model=cfg.embedder_compile, use_optimum=cfg.embedder_optimum, device=device, check=False)


def load_embedding_dataset(cfg, embedder: embedders.Embedder, *, use_targets: Optional[bool] = True, training: bool = False, strict_embedder: bool = True):
	"""Embedding-cache datasets only (the NounDataset branch of the reference builds its embeddings with the text tower first: out of scope here, write a cache)."""
	path = cfg.embedding_dataset
	if not path:
		raise ValueError("Cannot load embedding dataset given by empty string")
	if path.lower() == "noundataset":
		raise NotImplementedError("embedding_dataset=NounDataset needs the reference's noun dictionary tooling; train from an embedding cache file instead")
	if cfg.embedding_cache_dir and not os.path.isabs(path) and not os.path.exists(path) and os.path.exists(alt := os.path.join(resolve_source_path(cfg.embedding_cache_dir), path)):
		path = alt
	cache = embedding_cache.EmbeddingCache(path, embedder, use_targets=use_targets, strict_embedder=strict_embedder)
	return cache.create_dataset(batch_size=cfg.batch_size, training=training)


def gen_target_config(cfg, embedder: embedders.Embedder, targets: tuple, num_invalid_targets: int) -> embedders.TargetConfig:
	model_class = getattr(embedding_decoder, cfg.model)
	asked = dict(with_start_token=cfg.with_start_token, with_end_token=cfg.with_end_token, compact_ids=cfg.compact_ids, fixed_token_length=cfg.fixed_token_length,
	             auto_fixed_token_length=cfg.auto_fixed_token_length, use_masks=cfg.use_masks)
	kwargs = model_class.get_target_config_kwargs(**asked)
	if kwargs.keys() != asked.keys():
		raise ValueError("Model unexpectedly changed the target configuration keys")
	tc = embedder.create_target_config(targets=targets, **kwargs)
	embedder.configure_target(target_config=tc, target_vocab=targets[num_invalid_targets:] if num_invalid_targets > 0 else targets)
	return tc


def gen_data_config(cfg, dataset, **kwargs) -> embedding_dataset.DataConfig:
	model_class = getattr(embedding_decoder, cfg.model)
	asked = dict(use_weights=cfg.use_weights, unit_weights=None, multi_target=cfg.multi_target, multi_first=cfg.multi_first, full_targets=None,
	             fixed_multi_length=cfg.fixed_multi_length, multi_length=None)
	asked.update(kwargs)
	data_kwargs = model_class.get_data_config_kwargs(**asked)
	if set(data_kwargs) != {f.name for f in dataclasses.fields(embedding_dataset.DataConfig)}:
		raise ValueError("Model unexpectedly changed the data configuration keys or some keys are unexpected")
	dc = dataset.resolve_data_config(**data_kwargs)
	dataset.configure_data(dc)
	return dc


def load_decoder_checkpoint(cfg, hydra_dir: Optional[str] = None, checkpoint_path: Optional[str] = None, target_config=None, data_config=None, log: Callable[[str], None] = print):
	if checkpoint_path is None:
		checkpoint_path = cfg.load_model
		if not checkpoint_path:
			return None, None
	elif not checkpoint_path:
		raise ValueError("Cannot explicitly load a decoder checkpoint corresponding to an empty string")
	if hydra_dir is not None and not os.path.isabs(checkpoint_path) and not os.path.exists(checkpoint_path) and os.path.exists(alt := os.path.join(os.path.dirname(hydra_dir), checkpoint_path)):
		checkpoint_path = alt
	checkpoint_path = os.path.abspath(checkpoint_path)
	checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
	check_loaded_config("hydra config", _flat_cfg(cfg), checkpoint["cfg_flat"], ignore=IGNORE_CFG_DIFFS, log=log)
	if target_config is not None:
		check_loaded_config("target config", dataclasses.asdict(target_config), checkpoint["target_config"], log=log)
	if data_config is not None:
		check_loaded_config("data config", dataclasses.asdict(data_config), checkpoint["data_config"], log=log)
	return checkpoint, checkpoint_path


def action_train(cfg, hydra_dir: str, use_wandb: bool, *, log: Callable[[str], None] = print, on_chunk: Optional[Callable[[dict], None]] = None,
                 loader_seed: Optional[int] = None) -> dict[str, Any]:
	"""Decoder training from an embedding-cache file, driven by the reference's configuration keys (reference train.py:977-1190).

	Data parallel: run one process per GPU with torch.distributed initialised (backend 'nccl' = RCCL); every rank calls this with the same cfg.  Rank r reads batches
	r, r + world, ... of ONE shuffled order (seed `loader_seed`, default cfg.determ_seed), so `world` ranks with accum_factor a train the reference's accum_factor
	world * a.  Returns the objects of the run (model, optimizer, schedule, loop config / state) for callers that go on to evaluate.
	"""
	device, device_is_cpu, _ = infer.load_device(cfg.device)
	dp = DataParallel()
	embedder = load_embedder(cfg, device)
	dataset = load_embedding_dataset(cfg, embedder, use_targets=True, training=True, strict_embedder=cfg.strict_embedder)
	target_config = gen_target_config(cfg, embedder, dataset.targets, dataset.num_invalid_targets)
	data_config = gen_data_config(cfg, dataset)
	seed = loader_seed if loader_seed is not None else (int(cfg.determ_seed) if (cfg.determ or dp.enabled) else None)
	# (a STREAMING loader -- a cache beyond the HBM budget -- stages its batches through a ring of pinned + device slabs: two optimizer steps' worth of them, so that the
	# host can enqueue a whole step ahead of the device; with the default four slabs it ran in lockstep with the gathers of the step before: 0.71-0.84 of the bare step's rate)
	loader = embedding_cache.DeviceLoader(dataset, device, seed=seed, rank=dp.rank, world=dp.world, stream_depth=max(4, 2 * int(cfg.accum_factor) + 2),
	                                      group=int(cfg.accum_factor) if int(cfg.accum_factor) <= 32 else 1)  # (an optimizer step's micro-batches by one gather launch)
	loader_info = loader.loader_info
	grad_accum = embedding_dataset.GradAccum(loader=loader, loader_info=loader_info, accum_size=cfg.accum_factor, drop_last=True)
	C = make_train_loop_config(run_dir=hydra_dir, batch_size=grad_accum.batch_size, epoch_batches=grad_accum.loader_batches, num_valid_targets=dataset.num_valid_targets,
	                           accum_size=grad_accum.accum_size, chunk_scale=cfg.chunk_scale, max_chunks=cfg.max_chunks, max_epochs=cfg.max_epochs, save_every_min=cfg.save_every_min,
	                           save_every_max=cfg.save_every_max, save_top1_min=cfg.save_top1_min, save_top1_delta=cfg.save_top1_delta, gradient_clip=cfg.gradient_clip,
	                           loss_ewa_halflife=cfg.loss_ewa_halflife, last_dropout_chunks=cfg.last_dropout_chunks, last_dropout_factor=cfg.last_dropout_factor,
	                           use_wandb=use_wandb, device_is_cpu=device_is_cpu)
	log(f"Have {loader_info.available_samples} training samples available in the dataset{f' (rank {dp.rank} of {dp.world})' if dp.enabled else ''}")
	log(f"Training {grad_accum.loader_batches} batches = {grad_accum.loader_samples} samples per epoch (gradient accumulation factor {grad_accum.accum_size} => "
	    f"{grad_accum.loader_steps} optimizer updates), {C.chunk_batches} batches = {C.chunk_samples} samples per chunk, nominally {C.max_chunks} chunks")

	mean_shift = None
	if cfg.mean_shift:  # per-embedder modality-gap vector (reference :1008-1024, data/modality_gap/*.json)
		import json
		path = resolve_source_path(cfg.mean_shift_path.replace("$EMBEDDER", safe_embedder_spec(cfg.embedder_spec)))
		with open(path, "r") as f:
			js = json.load(f)
		for key, value in js.get("cfg_embedder", {}).items():
			if (mine := getattr(cfg, key, None)) != value:
				msg = f"Mean shift was calculated with {key}={value} but current config has {key}={mine}"
				if key == "embedder_spec":
					raise ValueError(msg)
				log("WARNING: " + msg)
		mean_shift = torch.tensor(js["mean_shift"], dtype=embedder.embed_dtype, device=device)
		if mean_shift.shape != (embedder.embed_dim,):
			raise ValueError(f"Mean shift has wrong shape: {tuple(mean_shift.shape)} vs {(embedder.embed_dim,)}")
		mean_shift = mean_shift.unsqueeze(0)
	embed_noise = embedding_noise.EmbeddingNoise.create(scheme=cfg.noise_scheme, embed_dim=embedder.embed_dim, vec_norm=cfg.noise_vec_norm, angle_min=cfg.noise_angle_min,
	                                                    angle_max=cfg.noise_angle_max, angle_std=cfg.noise_angle_std, mix_ratio=cfg.noise_mix_ratio)

	with dataset.loaded():
		checkpoint, checkpoint_path = load_decoder_checkpoint(cfg, hydra_dir=hydra_dir, target_config=target_config, data_config=data_config, log=log)
		if checkpoint is None:
			log("Training model from scratch")
		elif cfg.load_train_state:
			log(f"Resuming training from checkpoint: {checkpoint_path}")
			check_loaded_config("train loop config", dataclasses.asdict(C), checkpoint["train_loop_config"], ignore={"run_dir"}, log=log)
		else:
			log(f"Starting training from pretrained weights: {checkpoint_path}")
		model = infer.load_decoder_model(cfg, embedder, data_config, checkpoint)
		if checkpoint is not None and not cfg.load_train_state:
			checkpoint = None
		S = TrainLoopState() if checkpoint is None else utils.dataclass_from_dict(TrainLoopState, checkpoint["train_loop_state"])
		model.to(device)
		dp.broadcast_parameters(model.flat_parameters())  # every rank starts from rank 0's weights (identical anyway when seeded / loaded alike)
		model._shadow_version = -1
		if checkpoint is not None and cfg.load_lr_state:
			ck = checkpoint["cfg_flat"]
			init_lr, final_lr, lr_scheduler, lr_warmup = ck["init_lr"], ck["final_lr"], ck["lr_scheduler"], ck["lr_warmup"]
		else:
			init_lr, final_lr, lr_scheduler, lr_warmup = cfg.init_lr, cfg.final_lr, cfg.lr_scheduler, cfg.lr_warmup
		if cfg.optimizer.lower() != "adamw":
			raise ValueError(f"Unsupported optimizer: {cfg.optimizer} (the fused update is AdamW; the reference's AdamP option needs timm)")
		optimizer = FusedAdamW(model, lr=init_lr, betas=(cfg.beta1, cfg.beta2), weight_decay=cfg.weight_decay, max_norm=cfg.gradient_clip, weight_decay_1d=cfg.weight_decay_1d)
		if checkpoint is not None:
			if checkpoint["optimizer_type"] == "novic_amd.train.FusedAdamW":
				optimizer.load_state_dict(checkpoint["optimizer_state_dict"])
				if not cfg.load_lr_state:
					optimizer.param_groups[0]["lr"] = optimizer.param_groups[0]["initial_lr"] = init_lr
			elif checkpoint["optimizer_type"] == "torch.optim.adamw.AdamW":  # a `.train` file written by the reference itself (train.py:1464-1465)
				names = dict(model.named_parameters())
				optimizer.load_reference_state_dict(checkpoint["optimizer_state_dict"], [k for k in checkpoint["model_state_dict"] if k in names])
				log(f"Loaded the reference's torch.optim.AdamW state ({len(checkpoint['optimizer_state_dict']['state'])} tensors, step {optimizer.step_count}) into the fused optimizer")
				if not cfg.load_lr_state:
					optimizer.param_groups[0]["lr"] = optimizer.param_groups[0]["initial_lr"] = init_lr
			else:
				log(f"WARNING: loaded optimizer type ({checkpoint['optimizer_type']}) is not AdamW => not loading the optimizer state")
		# warm-up + cosine, stepped once per chunk; cosine horizon as the reference computes it at (re)start (:1154)
		t_max = max((C.max_chunks if final_lr > 0 else C.max_chunks + 1) - S.chunk_id, 1)
		schedule = ChunkSchedule(optimizer, init_lr, lr_warmup, lr_scheduler, t_max, final_lr) if (lr_scheduler.lower() != "const" or lr_warmup >= 1) else None
		if schedule is not None and checkpoint is not None and cfg.load_lr_state:
			sd_c, sd_w = checkpoint.get("scheduler_state_dict"), checkpoint.get("scheduler_warmup_state_dict")
			if sd_c and "chunks_done" in sd_c:
				schedule.load_state_dict(sd_c)
			elif (sd_c and "last_epoch" in sd_c) or (sd_w and "last_epoch" in sd_w):  # torch scheduler state dicts: the reference's own file
				schedule.load_reference_state_dicts(sd_w, sd_c)
		dp.decorrelate(model, embed_noise)
		if checkpoint is not None:
			rng = checkpoint.get("novic_rng_state") or {}
			model._dropout_calls = int(rng.get("dropout_calls", 0))
			if embed_noise is not None:
				embed_noise.calls = int(rng.get("noise_calls", 0))
			if "loader" in rng:
				loader.load_state_dict(rng["loader"])
		del checkpoint
		if cfg.dry_run:
			log("Dry run: not training")
		else:
			training_loop(_flat_cfg(cfg), C, S, model, dataset.targets, dataset.num_invalid_targets, mean_shift, embed_noise, grad_accum, optimizer, schedule, device, dp=dp, log=log,
			              on_chunk=on_chunk, rng_state=lambda: dict(loader=loader.state_dict()))
	return dict(model=model, optimizer=optimizer, schedule=schedule, train_loop_config=C, train_loop_state=S, embedder=embedder, dataset=dataset, loader=loader, noise=embed_noise)
