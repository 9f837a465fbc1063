"""Thin torch-tensor -> C-ABI adapters (one function per entry point of include/novic_hip.h).

PyTorch here is plumbing only: it owns device memory and the HIP stream; every FLOP below runs in libnovic_hip.so.
"""
from __future__ import annotations

import contextlib
import ctypes
import gc
import threading
from typing import Optional

import torch

from . import _lib
from ._lib import AdamWHyper, Epilogue, check

EPI_STORE_BF16, EPI_STORE_F32, EPI_ATOMIC_F32, EPI_RESID_F32, EPI_GELU_BF16, EPI_GELU_BWD_BF16, EPI_RESID_F16 = range(7)
ACT_NONE, ACT_GELU, ACT_QUICKGELU, ACT_GELU_TANH, ACT_RELU, ACT_TANH, ACT_IDENTITY = range(7)
ACT_BY_NAME = {"gelu": ACT_GELU, "relu": ACT_RELU, "tanh": ACT_TANH}  # the reference's utils.get_activation_gain names (utils.py:100-110)

_vp = ctypes.c_void_p


def _ptr(t: Optional[torch.Tensor]):
	return _vp(0) if t is None else _vp(t.data_ptr())


def _stream():
	return _vp(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
	for t in ts:
		if t is not None and not t.is_cuda:
			raise _lib.NovicHipError("novic_amd kernels need tensors on an MI355X device (no CPU fallback exists)")


class Dropout:
	"""Dropout descriptor shared by forward and backward of one site (mask = hash of (seed, site, element index), csrc/common.hpp)."""
	__slots__ = ("p", "seed", "site")

	def __init__(self, p: float = 0.0, seed: int = 0, site: int = 0):
		self.p, self.seed, self.site = float(p), int(seed), int(site)

	def at(self, site: int) -> "Dropout":
		return Dropout(self.p, self.seed, site)


NO_DROPOUT = Dropout()

# Workgroup budget of the persistent 256-wide GEMM grids, PER CALL (novic_epilogue_t.max_workgroups, ABI 8): `with ops.cu_budget(208): tower(images)` launches every GEMM
# inside on 208 CUs and leaves the library's process-wide default alone -- a thread-local of the HOST binding, so two host threads (a serving loop beside a trainer)
# cannot disturb each other the way flipping novic_persistent_cus between launches could.  0 / None = the library default.
_tls = threading.local()


class cu_budget:
	def __init__(self, n: Optional[int]):
		self.n = int(n) if n else 0
		if self.n and not 8 <= self.n <= 256:
			raise ValueError("cu_budget: 8..256 workgroups (0 / None = the library default)")

	def __enter__(self):
		self.prev = getattr(_tls, "cus", 0)
		_tls.cus = self.n
		return self

	def __exit__(self, *exc):
		_tls.cus = self.prev
		return False


# ---- streams of the package, created ONCE per device and in a fixed order -------------------------------------------------------------------------------------------
# The ROCm runtime maps HIP streams onto GPU_MAX_HW_QUEUES hardware queues in the order the process starts using them, and streams that share a queue run one after the
# other.  Three decode lanes reach 245-250 k greedy labels/s on queues of their own and 158 k when one of them shares (tools/lane_queue_probe.py: with 5 or 6 OTHER
# streams used before the lanes' own they collapse, with 0-4 or 7+ they do not) -- round 4's bench lost a third of its three-lane figures because a new leg in front of
# them, and a throw-away stream per graph capture, had shifted the lanes onto a shared queue.  So: the lane streams are reserved the first time anything of the package
# touches a device (they are streams number 0..3 of the process in the usual case), every graph capture runs on ONE capture stream, and nothing creates throw-away streams.
_STREAMS: dict = {}
# (Round 6 measured the 'tower' stream at a higher HIP priority than the decode beside it: four caller batches per launch 85.0 k -> 84.3 k labels/s, batch 1 024 on 208 CUs
# 83.3 k -> 84.4 k -- nothing; it only helped where the tower's grids take all 256 CUs, 76.2 k -> 81.4 k, which the workgroup budget already does better.  Not kept.)
N_LANE_STREAMS = 4
_NAMED_STREAMS = ("tower", "h2d", "loader", "wgrad")


def _device_streams(device) -> dict:
	dev = torch.device(device)
	if dev.index is None:
		dev = torch.device("cuda", torch.cuda.current_device())
	d = _STREAMS.get(dev)
	if d is None:
		d = _STREAMS[dev] = dict(lanes=[torch.cuda.Stream(device=dev) for _ in range(N_LANE_STREAMS)])
		for name in ("capture",) + _NAMED_STREAMS:
			d[name] = torch.cuda.Stream(device=dev)
		# (Which streams end up sharing a hardware queue: tools/stream_queue_map.py -- with eight queues the pairs that serialise are streams number (6, 7), (5, 8), (4, 9),
		# (3, 10), (2, 11) of the process.  A trivial launch on each stream in this order at reservation time, to pin the binding, made three lanes SLOWER (137 k, host-bound
		# graph launches): not kept; what is kept is the fixed creation order and the absence of throw-away streams.)
	return d


def lane_streams(device, n: int) -> list:
	"""The first n of the device's lane streams (decode lanes, tower lanes); more than N_LANE_STREAMS are created on demand behind them."""
	d = _device_streams(device)
	while len(d["lanes"]) < n:
		d["lanes"].append(torch.cuda.Stream(device=d["lanes"][0].device))
	return d["lanes"][:n]


@contextlib.contextmanager
def graph_capture(graph: "torch.cuda.CUDAGraph", stream: "torch.cuda.Stream"):
	"""`torch.cuda.graph(graph, stream=stream)` with Python's cyclic garbage collector OFF while the capture runs.  A collection that starts
	in the middle of a capture finalises whatever unreachable cycles earlier work left behind -- decode sessions with their graphs, events and pinned buffers hang in a
	cycle with their model -- and a destructor that makes a HIP call the capture forbids aborts the process (seen once: `Fatal Python error: Aborted`, `Garbage-collecting`
	under `_DecodeSession._capture`, round 5).  torch.cuda.graph collects once when it is entered (explicit collections work while the collector is disabled); nothing stops
	an automatic one DURING the capture but switching the collector off."""
	# Round 6, by construction rather than by habit:
	#  * ONE capture at a time in the process (`_capture_lock`, re-entrant for the thread that holds it) and a depth count, so the collector comes back on only when the
	#    LAST open capture of the package has ended -- two overlapping captures (a tower slot on one thread, a decode session on another) used to re-enable it under the
	#    other's feet;
	#  * `capture_error_mode="thread_local"`: torch's default ("global") makes a HIP call that a capture forbids -- the free of a pinned buffer, an event or a graph by
	#    reference count -- an error on EVERY thread while any capture is open; thread-local confines the check to the capturing thread, so a loader / stager thread
	#    (`novic-loader-stage`, `embedders.ImageStager` callers) that drops its last reference meanwhile neither aborts the process nor invalidates the capture
	#    (tests/test_gpu_generate.py::test_another_thread_may_free_hip_objects_while_a_capture_is_open).  The capturing thread itself allocates nothing page-locked and
	#    frees nothing inside a capture: its launches are the C ABI's, which neither allocate nor synchronise.
	#  * an EXPLICIT gc.collect() on another thread (the automatic collector is off process-wide) finalises whatever garbage the process holds -- old graphs with their
	#    memory pools among it -- and that did abort the process in a full-suite run of this round even under thread-local capture (no HIP error text: an abort inside the
	#    runtime).  `_gc_guard`, a `gc.callbacks` hook, makes such a collection WAIT for the capture to end (the capturing thread never waits for another thread's
	#    collection: nothing in the package joins a thread or takes a foreign lock inside a capture).
	#  * WHAT MUST NOT HAPPEN, measured (tools/capture_free_probe.py, one object kind and capture mode per process): of everything another thread may free while a capture
	#    is open -- pinned buffers, device tensors, events, streams -- only the destruction of a captured hipGraph (torch.cuda.CUDAGraph: graph + executable) takes the
	#    process down, in every capture mode, with no error text (an abort inside the runtime); a page-locked allocation on another thread is an error under "global" and
	#    fine under "thread_local".  So the package's graph holders (decode sessions, tower slots) never destroy a graph themselves: they hand it to `retire_graphs`, which
	#    destroys it at once when no capture is open (holding the capture lock, so that none can begin meanwhile) and parks it otherwise; the thread that closes the last
	#    capture empties the park.
	global _capture_depth, _capture_gc_was_on, _capture_owner
	with _capture_lock:
		if _capture_depth == 0:
			_capture_gc_was_on = gc.isenabled()
			gc.disable()
			_capture_owner = threading.get_ident()
		_capture_depth += 1
		try:
			with torch.cuda.graph(graph, stream=stream, capture_error_mode="thread_local"):
				yield
		finally:
			_capture_depth -= 1
			if _capture_depth == 0:
				_capture_owner = None
				_empty_park()  # (the capture has ended; the lock is still held: no other capture can begin under these destructors)
				if _capture_gc_was_on:
					gc.enable()


_capture_lock = threading.RLock()
_capture_depth = 0
_capture_gc_was_on = False
_capture_owner = None


_park_lock = threading.Lock()
_parked: list = []  # captured graphs whose holders died while a capture was open


def _empty_park():
	with _park_lock:
		dead = _parked[:]
		_parked.clear()
	del dead  # destroyed here


def retire_graphs(graphs):
	"""Destroy captured graphs (torch.cuda.CUDAGraph objects; `graphs`: a list, emptied) -- NOW if no capture of the package is open, else when the last open capture has
	ended.  Called by every holder of captured graphs when it lets go of them (`_DecodeSession.__del__`, `tower_runtime._Slot.__del__`, slot eviction): destroying a hipGraph
	while any thread is capturing aborts the process (see graph_capture)."""
	if not graphs:
		return
	held = list(graphs)
	try:
		graphs.clear()
	except (AttributeError, TypeError):
		pass
	if _capture_lock.acquire(blocking=False):
		try:
			if _capture_depth == 0:
				_empty_park()
				del held  # destroyed here, with the capture lock held: no capture can begin meanwhile
				return
			with _park_lock:  # (this very thread is capturing and dropped a holder: after its capture)
				_parked.extend(held)
		finally:
			_capture_lock.release()
	else:  # another thread is capturing: it empties the park when it closes its capture
		with _park_lock:
			_parked.extend(held)


def _gc_guard(phase, info):
	"""gc.callbacks hook: a collection that starts on a thread OTHER than the capturing one while a capture is open waits until the capture has ended."""
	if phase == "start" and _capture_depth > 0 and _capture_owner is not None and _capture_owner != threading.get_ident():
		with _capture_lock:
			pass


if _gc_guard not in gc.callbacks:
	gc.callbacks.append(_gc_guard)


def capture_open() -> bool:
	"""True while a hipGraph capture of this package is open on any thread (diagnostics / tests)."""
	return _capture_depth > 0


def capture_stream(device) -> "torch.cuda.Stream":
	"""The one stream hipGraph captures of this package run on (a capture needs a non-default stream; a fresh one per capture would use up the runtime's queue slots)."""
	return _device_streams(device)["capture"]


def named_stream(device, name: str) -> "torch.cuda.Stream":
	"""One stream per (device, role): 'tower' (pipelined image towers), 'h2d' (image staging), 'loader' (streaming cache loader), 'wgrad'."""
	d = _device_streams(device)
	st = d.get(name)
	if st is None:  # (a role outside the fixed set: created behind it)
		st = d[name] = torch.cuda.Stream(device=d["capture"].device)
	return st


def set_cu_budget(n: Optional[int]) -> int:
	"""The non-scoped form of cu_budget (a backward pass that starts its early all-reduces half way through): sets this thread's budget, returns the previous one."""
	prev = getattr(_tls, "cus", 0)
	_tls.cus = int(n) if n else 0
	return prev


def current_cu_budget() -> int:
	"""The workgroups a gemm() issued now may have: the enclosing cu_budget, else the library default (novic_persistent_cus / $NOVIC_PERSISTENT_CUS)."""
	n = getattr(_tls, "cus", 0)
	return n // 8 * 8 if n else persistent_cus()


def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, a_kstrided=False, b_kstrided=False, kind=EPI_STORE_BF16, out: torch.Tensor,
         out2: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act=ACT_NONE,
         alpha: float = 1.0, split_k: int = 1, dropout: Dropout = NO_DROPOUT, lda: Optional[int] = None, ldb: Optional[int] = None,
         ldc: Optional[int] = None, ldr: Optional[int] = None, row_limit: Optional[torch.Tensor] = None, split_tail: bool = False):
	"""C[M,N] = A*B with a fused epilogue (novic_gemm_bf16).  a/b are bf16 2-D tensors in the storage the flags name.
	row_limit: optional device int32 scalar -- only the first row_limit token rows take part (M, or K for the weight-gradient form).
	split_tail: hand the kernel this device's K-split scratch (novic_epilogue_t.splitk_ws): the output tiles behind the last whole round of 256 are
	cut along K -- deterministic, but not bit-identical to the unsplit kernels (the ViT / text towers ask for it; calls must share one stream).
	out2: GELU_BF16's second output (the bf16 pre-activation)."""
	_dev(a, b, out)
	assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
	ep = Epilogue()
	ep.struct_bytes = ctypes.sizeof(Epilogue)
	ep.kind, ep.act = kind, act
	ep.c, ep.c2, ep.resid, ep.bias = out.data_ptr(), (out2.data_ptr() if out2 is not None else 0), (resid.data_ptr() if resid is not None else 0), (bias.data_ptr() if bias is not None else 0)
	ep.ldc = ldc if ldc is not None else out.stride(-2)
	ep.ldr = ldr if ldr is not None else (resid.stride(-2) if resid is not None else 0)
	ep.alpha, ep.drop_p = alpha, dropout.p
	ep.seed_lo, ep.seed_hi, ep.drop_site = dropout.seed & 0xFFFFFFFF, (dropout.seed >> 32) & 0xFFFFFFFF, dropout.site
	ep.row_limit = row_limit.data_ptr() if row_limit is not None else 0
	ep.max_workgroups = getattr(_tls, "cus", 0)
	if split_tail:
		ws = _splitk_ws(out.device)
		ep.splitk_ws, ep.splitk_ws_bytes = ws.data_ptr(), ws.numel() * 4
	rc = _lib.lib().novic_gemm_bf16(_ptr(a), _ptr(b), M, N, K, lda if lda is not None else a.stride(-2), ldb if ldb is not None else b.stride(-2),
	                                int(a_kstrided), int(b_kstrided), split_k, ctypes.byref(ep), _stream())
	check(rc, "novic_gemm_bf16")
	return out


def wgrad(dy: torch.Tensor, x: torch.Tensor, M: int, N: int, K: int, out: torch.Tensor, *, alpha: float = 1.0, row_limit: Optional[torch.Tensor] = None, splits: int = 0):
	"""out[M][N] (fp32) += alpha * dy[:K, :M]^T x[:K, :N] (novic_wgrad_bf16: 256 x 256 tiles, fixed-order partial sums through this device's scratch).  Inside
	`cu_budget(n)` the launch takes at most n workgroups (the part count follows: another fp32 summation order than without a budget, deterministic for a given n)."""
	_dev(dy, x, out)
	assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and out.dtype == torch.float32
	ws = _splitk_ws(out.device)
	check(_lib.lib().novic_wgrad_bf16(_ptr(dy), _ptr(x), M, N, K, dy.stride(0), x.stride(0), _ptr(out), out.stride(0), ctypes.c_float(alpha), _ptr(row_limit), _ptr(ws),
	                                  ctypes.c_uint64(ws.numel() * 4), int(splits), int(getattr(_tls, "cus", 0)), _stream()), "novic_wgrad_bf16")
	return out


def wgrad2(dy1: torch.Tensor, x1: torch.Tensor, M1: int, N1: int, out1: torch.Tensor, dy2: torch.Tensor, x2: torch.Tensor, M2: int, N2: int, out2: torch.Tensor, K: int, *,
           alpha: float = 1.0, row_limit: Optional[torch.Tensor] = None):
	"""Two weight gradients over the same K token rows in one launch pair (novic_wgrad2_bf16): out_i[M_i][N_i] += alpha * dy_i[:K, :M_i]^T x_i[:K, :N_i]."""
	_dev(dy1, x1, out1, dy2, x2, out2)
	ws = _splitk_ws(out1.device)
	check(_lib.lib().novic_wgrad2_bf16(_ptr(dy1), _ptr(x1), M1, N1, dy1.stride(0), x1.stride(0), _ptr(out1), out1.stride(0), _ptr(dy2), _ptr(x2), M2, N2, dy2.stride(0),
	                                   x2.stride(0), _ptr(out2), out2.stride(0), K, ctypes.c_float(alpha), _ptr(row_limit), _ptr(ws), _u64(ws.numel() * 4),
	                                   int(getattr(_tls, "cus", 0)), _stream()), "novic_wgrad2_bf16")


def colsum_bf16(x: torch.Tensor, rows: int, cols: int, out: torch.Tensor, *, alpha: float = 1.0, row_limit: Optional[torch.Tensor] = None):
	"""out[:cols] (fp32) += alpha * x[:rows (device row_limit), :cols].sum(0) (novic_colsum_bf16: a bias gradient; deterministic, partial sums through this device's scratch)."""
	_dev(x, out)
	ws = _splitk_ws(out.device)
	check(_lib.lib().novic_colsum_bf16(_ptr(x), rows, cols, x.stride(0), _ptr(row_limit), _ptr(out), ctypes.c_float(alpha), _ptr(ws), _u64(ws.numel() * 4), _stream()), "novic_colsum_bf16")


def wgrad_policy(policy: int = -1) -> int:
	"""1: the 8-phase weight-gradient kernel (default), 0: the one-barrier-per-K-tile kernel; returns the previous policy (-1 only queries)."""
	return int(_lib.lib().novic_wgrad_policy(int(policy)))


def wgrad_supported(M: int, N: int, K: int) -> bool:
	"""Shapes the 256-wide weight-gradient kernel is meant for: many output tiles, a long token dimension (else the 64 MiB of partial sums outweigh the operands)."""
	if M % 8 or N % 8 or K < 16384:
		return False
	if min(M, N) <= 128:  # the feed-forward gradients [128 x 512] / [512 x 128] CAN run on 128 x 256 tiles (the narrow dimension as the tile rows), but inside the
		return False      # step the kernel + its reduction take 27 + 12 us against 35 us of the 128^2 split-K kernel: not taken (tools/wgrad_bench.py)
	tiles = ((M + 255) // 256) * ((N + 255) // 256)
	return 4 <= tiles <= 256  # measured: in-proj dW 150 -> 108 us, logits dW 414 -> 273, out-proj dW 62 -> 53


_SPLITK_WS: dict = {}


def _splitk_ws(device) -> torch.Tensor:
	"""64 MiB of fp32 scratch for the K-split tail tiles / weight-gradient partial sums (256 partial 256 x 256 accumulator tiles at most), one per (device, stream):
	launches on different streams run concurrently (the weight gradients of overlap_wgrad on their side stream beside a split_tail GEMM on the main stream) and would
	overwrite each other's partial sums in a shared buffer; launches on ONE stream are ordered, so they can share."""
	own = getattr(_tls, "splitk", None)
	if own is not None:  # `with splitk_scratch(t):` -- a scratch the caller owns (the towers: one per batch-shape slot and lane, freed with the slot's captured graph)
		return own
	dev = torch.device(device)
	key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
	ws = _SPLITK_WS.get(key)
	if ws is None:
		ws = _SPLITK_WS[key] = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=device)
	return ws


class splitk_scratch:
	"""`with ops.splitk_scratch(t):` -- gemm(split_tail=True) / wgrad* inside use the fp32 tensor t (>= 64 MiB) instead of the per-(device, stream) scratch.  For launch
	sequences that are CAPTURED: the scratch baked into a hipGraph must be the graph owner's, not a global keyed by whichever throw-away stream the capture ran on (torch
	recycles stream handles: two graphs, or a graph and an eager lane, could end up sharing one scratch while running concurrently)."""

	def __init__(self, t: torch.Tensor):
		assert t.dtype == torch.float32 and t.is_cuda and t.numel() >= 16 * 1024 * 1024
		self.t = t

	def __enter__(self):
		self.prev = getattr(_tls, "splitk", None)
		_tls.splitk = self.t
		return self.t

	def __exit__(self, *exc):
		_tls.splitk = self.prev
		return False


def _u64(x: int):
	return ctypes.c_uint64(x & 0xFFFFFFFFFFFFFFFF)


def _tok_bytes(t: torch.Tensor) -> int:
	if t.dtype == torch.int64:
		return 8
	if t.dtype == torch.int32:
		return 4
	raise TypeError(f"token ids must be int64 or int32, got {t.dtype}")


def rownorm_bf16(x: torch.Tensor, out: torch.Tensor):
	_dev(x, out)
	check(_lib.lib().novic_rownorm_bf16(_ptr(x), _ptr(out), x.shape[0], x.shape[1], out.stride(0), _stream()), "novic_rownorm_bf16")
	return out


def layernorm_fwd(x: torch.Tensor, gamma: torch.Tensor, out_bf16: Optional[torch.Tensor], rows_out: int, E: int, *, beta=None, out_f32=None, seq_in=1, seq_out=1,
                  seq_off=0, eps=1e-5):
	"""x fp32, or fp16 (the residual stream of a half-precision tower: novic_layernorm_fwd_f16, bf16 output only)."""
	_dev(x, gamma)
	if x.dtype == torch.float16:
		assert out_f32 is None and out_bf16 is not None
		check(_lib.lib().novic_layernorm_fwd_f16(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out_bf16), rows_out, E, seq_in, seq_out, seq_off, ctypes.c_float(eps), _stream()),
		      "novic_layernorm_fwd_f16")
		return
	assert x.dtype == torch.float32
	check(_lib.lib().novic_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out_bf16), _ptr(out_f32), rows_out, E, seq_in, seq_out, seq_off, ctypes.c_float(eps),
	                                     _stream()), "novic_layernorm_fwd")


def layernorm_fwd_rows(x: torch.Tensor, gamma: torch.Tensor, out_bf16: torch.Tensor, src_rows: Optional[torch.Tensor], row_count: torch.Tensor, rows_max: int, E: int, *,
                       beta=None, eps=1e-5):
	"""out[j] = LayerNorm(x[src_rows[j]]) (x[j] when src_rows is None) for j < row_count (device int32 scalar), at most rows_max rows."""
	_dev(x, gamma, out_bf16, row_count)
	check(_lib.lib().novic_layernorm_fwd_rows(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(out_bf16), _ptr(src_rows), _ptr(row_count), rows_max, E, ctypes.c_float(eps),
	                                          _stream()), "novic_layernorm_fwd_rows")


def layernorm_bwd(dy: torch.Tensor, x: torch.Tensor, gamma: torch.Tensor, dx_in: Optional[torch.Tensor], dx_out: torch.Tensor, g_out: Optional[torch.Tensor],
                  dgamma: Optional[torch.Tensor], rows_in: int, E: int, *, seq_in=1, seq_out=1, seq_off=0, eps=1e-5, dropout: Dropout = NO_DROPOUT,
                  dy_row: Optional[torch.Tensor] = None, row_limit: Optional[torch.Tensor] = None):
	"""dy_row (int32 [rows_in], optional): the upstream gradient of input row m is row dy_row[m] of dy (< 0: none) instead of the seq window.
	row_limit (device int32 scalar, optional): only the first row_limit input rows exist (packed rows)."""
	_dev(dy, x, dx_out)
	check(_lib.lib().novic_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(dx_in), _ptr(dx_out), _ptr(g_out), _ptr(dgamma), rows_in, E, seq_in, seq_out, seq_off,
	                                     ctypes.c_float(eps), ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(dropout.site), _ptr(dy_row), _ptr(row_limit), _stream()),
	      "novic_layernorm_bwd")


def hidden_norm_act_fwd(h0: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], out: torch.Tensor, rows: int, H: int, act: int, eps: float = 1e-5):
	"""out (bf16) = act(LayerNorm(h0 (bf16); gamma, beta)): the normalised hidden layer of the prefix MLP (novic_hidden_norm_act_fwd)."""
	_dev(h0, gamma, out)
	check(_lib.lib().novic_hidden_norm_act_fwd(_ptr(h0), _ptr(gamma), _ptr(beta), _ptr(out), rows, H, h0.stride(0), out.stride(0), int(act), ctypes.c_float(eps), _stream()),
	      "novic_hidden_norm_act_fwd")


def hidden_norm_act_bwd(dy: torch.Tensor, h0: torch.Tensor, gamma: torch.Tensor, beta: Optional[torch.Tensor], dh0: torch.Tensor, dgamma: torch.Tensor,
                        dbeta: Optional[torch.Tensor], rows: int, H: int, act: int, eps: float = 1e-5):
	"""dh0 (bf16) = LayerNorm'(dy * act'(z)); dgamma / dbeta accumulate (novic_hidden_norm_act_bwd)."""
	_dev(dy, h0, gamma, dh0, dgamma)
	check(_lib.lib().novic_hidden_norm_act_bwd(_ptr(dy), _ptr(h0), _ptr(gamma), _ptr(beta), _ptr(dh0), _ptr(dgamma), _ptr(dbeta), rows, H, dy.stride(0), h0.stride(0),
	                                           dh0.stride(0), int(act), ctypes.c_float(eps), _stream()), "novic_hidden_norm_act_bwd")


def layernorm_bwd_sum(dy16: Optional[torch.Tensor], dy32: Optional[torch.Tensor], x: torch.Tensor, gamma: torch.Tensor, dx_out: torch.Tensor, g_out: Optional[torch.Tensor],
                      dgamma: Optional[torch.Tensor], dbeta: Optional[torch.Tensor], rows: int, E: int, *, eps=1e-5, dropout: Dropout = NO_DROPOUT,
                      row_limit: Optional[torch.Tensor] = None):
	"""LayerNorm backward with the upstream gradient dy16 (bf16) + dy32 (fp32), either optional: the norms of post-LN layers (novic_layernorm_bwd_sum)."""
	_dev(x, gamma, dx_out)
	check(_lib.lib().novic_layernorm_bwd_sum(_ptr(dy16), _ptr(dy32), _ptr(x), _ptr(gamma), _ptr(dx_out), _ptr(g_out), _ptr(dgamma), _ptr(dbeta), rows, E, ctypes.c_float(eps),
	                                         ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(dropout.site), _ptr(row_limit), _stream()), "novic_layernorm_bwd_sum")


def add_bf16(dst: torch.Tensor, src: torch.Tensor):
	"""dst (fp32) += src (bf16), elementwise (novic_add_bf16)."""
	_dev(dst, src)
	assert dst.dtype == torch.float32 and src.dtype == torch.bfloat16 and dst.numel() == src.numel() and dst.is_contiguous() and src.is_contiguous()
	check(_lib.lib().novic_add_bf16(_ptr(dst), _ptr(src), _u64(dst.numel()), _stream()), "novic_add_bf16")


def rezero_fwd(resid: torch.Tensor, branch: torch.Tensor, scale: torch.Tensor, out: torch.Tensor, rows: int, E: int, row_limit: Optional[torch.Tensor] = None):
	"""out = resid + bf16(scale * branch) with a device scalar `scale` (novic_rezero_fwd: ReZero's `x *= scale` in front of the residual add)."""
	_dev(resid, branch, scale, out)
	check(_lib.lib().novic_rezero_fwd(_ptr(resid), _ptr(branch), _ptr(scale), _ptr(out), rows, E, _ptr(row_limit), _stream()), "novic_rezero_fwd")


def rezero_bwd(dx: torch.Tensor, branch: torch.Tensor, scale: torch.Tensor, dscale: torch.Tensor, g_out: torch.Tensor, rows: int, E: int, dropout: Dropout = NO_DROPOUT,
               row_limit: Optional[torch.Tensor] = None):
	"""dscale += sum bf16(dx) * branch; g_out = bf16(bf16(bf16(dx) * scale) * dropout mask) (novic_rezero_bwd)."""
	_dev(dx, branch, scale, dscale, g_out)
	check(_lib.lib().novic_rezero_bwd(_ptr(dx), _ptr(branch), _ptr(scale), _ptr(dscale), _ptr(g_out), rows, E, ctypes.c_float(dropout.p), _u64(dropout.seed),
	                                  ctypes.c_uint32(dropout.site), _ptr(row_limit), _stream()), "novic_rezero_bwd")


def embed_fwd(prefix: torch.Tensor, tokens: Optional[torch.Tensor], tok_ld: int, wtok: torch.Tensor, pos: torch.Tensor, x0: torch.Tensor, A, S, P, E, V, B, mrep,
              multi_first, dropout: Dropout = NO_DROPOUT, seq=None):
	"""seq = (seq_start, seq_len) int32 [A] device tensors (packed rows, seq_layout) or None (dense [A][S])."""
	_dev(prefix, wtok, pos, x0)
	tb = _tok_bytes(tokens) if tokens is not None else 8
	check(_lib.lib().novic_embed_fwd(_ptr(prefix), _ptr(tokens), tb, tok_ld, _ptr(wtok), _ptr(pos), _ptr(x0), A, S, P, E, V, B, mrep, int(multi_first),
	                                 ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(dropout.site), _ptr(seq[0] if seq else None),
	                                 _ptr(seq[1] if seq else None), _stream()), "novic_embed_fwd")


def embed_fwd_ln(prefix: torch.Tensor, tokens: Optional[torch.Tensor], tok_ld: int, wtok: torch.Tensor, pos: torch.Tensor, x0: torch.Tensor, A, S, P, E, V, B, mrep,
                 multi_first, gamma: torch.Tensor, ln_out: torch.Tensor, dropout: Dropout = NO_DROPOUT, seq=None, eps: float = 1e-5):
	"""embed_fwd + layer 0's norm1 of the rows it has just assembled (novic_embed_fwd_ln): x0 and ln_out = bf16(LayerNorm(x0; gamma)) in one launch."""
	_dev(prefix, wtok, pos, x0, gamma, ln_out)
	tb = _tok_bytes(tokens) if tokens is not None else 8
	check(_lib.lib().novic_embed_fwd_ln(_ptr(prefix), _ptr(tokens), tb, tok_ld, _ptr(wtok), _ptr(pos), _ptr(x0), A, S, P, E, V, B, mrep, int(multi_first),
	                                    ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(dropout.site), _ptr(seq[0] if seq else None),
	                                    _ptr(seq[1] if seq else None), _ptr(gamma), _ptr(ln_out), ctypes.c_float(eps), _stream()), "novic_embed_fwd_ln")


def ln_embed_bwd(dy: torch.Tensor, x0: torch.Tensor, gamma: torch.Tensor, dx_in: torch.Tensor, dgamma: torch.Tensor, tokens: Optional[torch.Tensor], tok_ld: int,
                 dwtok: torch.Tensor, dpos: torch.Tensor, dprefix: torch.Tensor, A, S, P, E, V, B, mrep, multi_first, dropout: Dropout = NO_DROPOUT, seq=None, eps: float = 1e-5):
	"""layer 0's norm1 backward in front of embed_bwd, one launch (novic_ln_embed_bwd): dx0 = dx_in + LN'(dy) is scattered from registers, never written."""
	_dev(dy, x0, gamma, dx_in, dgamma, dwtok, dpos, dprefix)
	tb = _tok_bytes(tokens) if tokens is not None else 8
	check(_lib.lib().novic_ln_embed_bwd(_ptr(dy), _ptr(x0), _ptr(gamma), _ptr(dx_in), _ptr(dgamma), _ptr(tokens), tb, tok_ld, _ptr(dwtok), _ptr(dpos), _ptr(dprefix), A, S, P, E,
	                                    V, B, mrep, int(multi_first), ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(dropout.site),
	                                    _ptr(seq[0] if seq else None), _ptr(seq[1] if seq else None), ctypes.c_float(eps), _stream()), "novic_ln_embed_bwd")


def embed_bwd(dx0: torch.Tensor, tokens: Optional[torch.Tensor], tok_ld: int, dwtok: torch.Tensor, dpos: torch.Tensor, dprefix: torch.Tensor, A, S, P, E, V, B, mrep,
              multi_first, dropout: Dropout = NO_DROPOUT, seq=None):
	_dev(dx0, dwtok, dpos, dprefix)
	tb = _tok_bytes(tokens) if tokens is not None else 8
	check(_lib.lib().novic_embed_bwd(_ptr(dx0), _ptr(tokens), tb, tok_ld, _ptr(dwtok), _ptr(dpos), _ptr(dprefix), A, S, P, E, V, B, mrep, int(multi_first),
	                                 ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(dropout.site), _ptr(seq[0] if seq else None),
	                                 _ptr(seq[1] if seq else None), _stream()), "novic_embed_bwd")


def dec_attn_fwd(qkv: torch.Tensor, key_pad: Optional[torch.Tensor], o: torch.Tensor, A, S, H, D, P, strict: bool, dropout: Dropout = NO_DROPOUT, seq=None):
	_dev(qkv, o)
	check(_lib.lib().novic_dec_attn_fwd(_ptr(qkv), _ptr(key_pad), _ptr(o), A, S, H, D, P, int(strict), ctypes.c_float(dropout.p), _u64(dropout.seed),
	                                    ctypes.c_uint32(dropout.site), _ptr(seq[0] if seq else None), _ptr(seq[1] if seq else None), _stream()), "novic_dec_attn_fwd")


def dec_attn_bwd(qkv: torch.Tensor, key_pad: Optional[torch.Tensor], d_o: torch.Tensor, dqkv: torch.Tensor, A, S, H, D, P, strict: bool, dropout: Dropout = NO_DROPOUT,
                 seq=None):
	_dev(qkv, d_o, dqkv)
	check(_lib.lib().novic_dec_attn_bwd(_ptr(qkv), _ptr(key_pad), _ptr(d_o), _ptr(dqkv), A, S, H, D, P, int(strict), ctypes.c_float(dropout.p), _u64(dropout.seed),
	                                    ctypes.c_uint32(dropout.site), _ptr(seq[0] if seq else None), _ptr(seq[1] if seq else None), _stream()), "novic_dec_attn_bwd")


def build_padding(target_padding: Optional[torch.Tensor], weight: Optional[torch.Tensor], key_pad: torch.Tensor, out_pad: torch.Tensor, A, C, P, num_end_loss, tpad_ld=None):
	_dev(key_pad, out_pad)
	check(_lib.lib().novic_build_padding(_ptr(target_padding), tpad_ld if tpad_ld is not None else C, _ptr(weight), _ptr(key_pad), _ptr(out_pad), A, C, P, num_end_loss,
	                                     _stream()), "novic_build_padding")


def cross_entropy(logits: torch.Tensor, ldl, V, A, T, C, col0, target: Optional[torch.Tensor], out_pad, weight, basis, group_rows, grad_scale, smoothing, write_grad,
                  row_loss, row_argmax, row_correct, argmax_from=0, grad_scale_dev=None, tok_ld=None, row_map=None, row_limit=None):
	"""row_map / row_limit (device int32, optional): compacted logits -- logits row j is token row row_map[j], j < row_limit (compact_rows)."""
	_dev(logits, row_loss, row_argmax)
	tb = _tok_bytes(target) if target is not None else 8
	check(_lib.lib().novic_cross_entropy(_ptr(logits), ldl, V, A, T, C, col0, _ptr(target), tb, (tok_ld if tok_ld is not None else C), _ptr(out_pad), _ptr(weight), _ptr(basis), group_rows,
	                                     ctypes.c_float(grad_scale), _ptr(grad_scale_dev), ctypes.c_float(smoothing), int(write_grad), _ptr(row_loss), _ptr(row_argmax), _ptr(row_correct),
	                                     argmax_from, _ptr(row_map), _ptr(row_limit), _stream()), "novic_cross_entropy")


def seq_layout(key_pad: torch.Tensor, A: int, S: int, seq_start: torch.Tensor, seq_len: torch.Tensor, total: torch.Tensor):
	"""Packed-row layout from the key padding: see novic_seq_layout.  total: int32 [1 + ceil(A/1024)], total[0] = the row count."""
	_dev(key_pad, seq_start, seq_len, total)
	assert total.numel() >= 1 + (A + 1023) // 1024
	check(_lib.lib().novic_seq_layout(_ptr(key_pad), A, S, _ptr(seq_start), _ptr(seq_len), _ptr(total), _stream()), "novic_seq_layout")


def compact_rows(out_pad, weight, A, T, C, col0, S, rows, src_rows, dst_of, count, row_loss=None, row_argmax=None, row_correct=None, seq_start=None):
	"""Lists the output positions that count (not padded, weight != 0) first: see novic_compact_rows.  count: int32 [1 + ceil(A*T/1024)], count[0] = the number."""
	_dev(rows, src_rows, dst_of, count)
	assert count.numel() >= 1 + (A * T + 1023) // 1024
	check(_lib.lib().novic_compact_rows(_ptr(out_pad), _ptr(weight), A, T, C, col0, S, _ptr(rows), _ptr(src_rows), _ptr(dst_of), _ptr(count), _ptr(row_loss), _ptr(row_argmax),
	                                    _ptr(row_correct), _ptr(seq_start), _stream()), "novic_compact_rows")


def loss_group_reduce(row_loss, row_correct, out_pad, weight, basis, loss, correct, tokens, A, T, C, col0, group_rows):
	check(_lib.lib().novic_loss_group_reduce(_ptr(row_loss), _ptr(row_correct), _ptr(out_pad), _ptr(weight), _ptr(basis), _ptr(loss), _ptr(correct), _ptr(tokens), A, T, C,
	                                         col0, group_rows, _stream()), "novic_loss_group_reduce")


NOISE_NONE, NOISE_GAUSS_ELEM, NOISE_GAUSS_VEC, NOISE_GAUSS_ANGLE, NOISE_UNIFORM_ANGLE, NOISE_GAUSS_ELEM_UNIFORM_ANGLE = range(6)


def noise_fused(embed: torch.Tensor, mode: int, *, vec_norm=0.0, angle_min=0.0, angle_max=0.0, angle_std=0.0, mix_ratio=0.0, seed=0, offset=0, inj_z1=None, inj_z2=None,
                inj_row=None, inj_mix=None, mean_shift=None):
	"""In place on B x F fp32 rows; angles in radians."""
	_dev(embed)
	assert embed.dtype == torch.float32 and embed.is_contiguous() and embed.ndim == 2
	f = ctypes.c_float
	check(_lib.lib().novic_noise_fused(_ptr(embed), embed.shape[0], embed.shape[1], mode, f(vec_norm), f(angle_min), f(angle_max), f(angle_std), f(mix_ratio), _u64(seed),
	                                   ctypes.c_uint32(offset & 0xFFFFFFFF), _ptr(inj_z1), _ptr(inj_z2), _ptr(inj_row), _ptr(inj_mix), _ptr(mean_shift), _stream()),
	      "novic_noise_fused")
	return embed


def grad_norm(grads: torch.Tensor, partial_ws: torch.Tensor, out_norm: torch.Tensor):
	check(_lib.lib().novic_grad_norm(_ptr(grads), ctypes.c_uint64(grads.numel()), _ptr(partial_ws), partial_ws.numel(), _ptr(out_norm), _stream()), "novic_grad_norm")


def adamw_step(params, grads, exp_avg, exp_avg_sq, shadow_bf16, n_decay: int, grad_norm_t: Optional[torch.Tensor], *, lr: float, beta1: float, beta2: float, eps: float,
               weight_decay: float, step: int, max_norm: float):
	"""step = 1-based optimizer step (bias corrections are formed here, in double precision).  The hyper-parameters go to the kernel by value."""
	_dev(params, grads, exp_avg, exp_avg_sq)
	assert step >= 1
	h = AdamWHyper(lr, beta1, beta2, eps, weight_decay, 1.0 - beta1 ** step, 1.0 - beta2 ** step, max_norm)
	check(_lib.lib().novic_adamw_step(_ptr(params), _ptr(grads), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(shadow_bf16), ctypes.c_uint64(params.numel()),
	                                  ctypes.c_uint64(n_decay), ctypes.byref(h), _ptr(grad_norm_t), _stream()), "novic_adamw_step")


def wgradn(problems, K: int, *, alpha: float = 1.0, row_limit: Optional[torch.Tensor] = None):
	"""Up to four weight gradients over the same K token rows in ONE launch pair (novic_wgradn_bf16): problems = [(dy, x, M, N, out), ...] with out_i[M_i][N_i] (fp32) +=
	alpha * dy_i[:K, :M_i]^T x_i[:K, :N_i] -- the attention pairs of two layers (32 tiles x 8 parts) or their narrow feed-forward pairs (8 tiles x 32 parts): half the
	partial-sum traffic of one novic_wgrad2_bf16 call per layer."""
	arr = (_lib.WgradProblem * len(problems))()
	for q, (dy, x, M, N, out) in zip(arr, problems):
		_dev(dy, x, out)
		assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and out.dtype == torch.float32
		q.dY, q.X, q.dW, q.M, q.N, q.ldy, q.ldx, q.ldw = dy.data_ptr(), x.data_ptr(), out.data_ptr(), int(M), int(N), dy.stride(0), x.stride(0), out.stride(0)
	ws = _splitk_ws(problems[0][4].device)
	check(_lib.lib().novic_wgradn_bf16(arr, len(problems), int(K), ctypes.c_float(alpha), _ptr(row_limit), _ptr(ws), _u64(ws.numel() * 4), int(getattr(_tls, "cus", 0)), _stream()),
	      "novic_wgradn_bf16")


def transpose_bf16_batched(src: torch.Tensor, dst: torch.Tensor, desc):
	"""desc: iterable of (src_off, dst_off, rows, cols[, dst_ld]) in elements; destination i is the transpose [cols][dst_ld >= rows] of source i."""
	_dev(src, dst)
	flat = [int(v) for d in desc for v in (tuple(d) if len(d) == 5 else tuple(d) + (d[2],))]
	arr = (ctypes.c_longlong * len(flat))(*flat)
	check(_lib.lib().novic_transpose_bf16_batched(_ptr(src), _ptr(dst), arr, len(flat) // 5, _stream()), "novic_transpose_bf16_batched")


def cast_bf16(x: torch.Tensor, y: torch.Tensor):
	check(_lib.lib().novic_cast_bf16(_ptr(x), _ptr(y), ctypes.c_uint64(x.numel()), _stream()), "novic_cast_bf16")


def _next_embed(x_next, wtok, pos_row, origin_in=None, origin_out=None, npos=0):
	"""novic_next_embed_t for a step kernel, or None"""
	if x_next is None:
		return None
	_dev(x_next, wtok, pos_row)
	ne = _lib.NextEmbed()
	ne.struct_bytes, ne.E = ctypes.sizeof(_lib.NextEmbed), x_next.shape[-1]
	ne.wtok, ne.pos_row, ne.x_next = wtok.data_ptr(), pos_row.data_ptr(), x_next.data_ptr()
	ne.origin_in = origin_in.data_ptr() if origin_in is not None else None
	ne.origin_out = origin_out.data_ptr() if origin_out is not None else None
	ne.npos = int(npos)
	return ne


def greedy_step(logits, ldl, V, B, G, step, ids, pad, alive, score, nll, count, active, step_logits, temperature, smoothing, x_next=None, wtok=None, pos_row=None):
	"""x_next (+ wtok, pos_row): also write the next step's input rows W_tok[chosen token] + pos_row (what novic_decode_embed would do in a launch of its own)."""
	ne = _next_embed(x_next, wtok, pos_row)
	check(_lib.lib().novic_greedy_step_next(_ptr(logits), ldl, V, B, G, step, _ptr(ids), _tok_bytes(ids), _ptr(pad), _ptr(alive), _ptr(score), _ptr(nll), _ptr(count), _ptr(active),
	                                        _ptr(step_logits), ctypes.c_float(temperature), ctypes.c_float(smoothing), ctypes.byref(ne) if ne is not None else None, _stream()),
	      "novic_greedy_step")


def host_mapped_ptr(t: torch.Tensor) -> int:
	"""Device address of a pinned host tensor's storage (novic_host_mapped_ptr: page-locked and mapped, else NovicHipError).  Call it when the buffer is made, never inside a capture."""
	if t.is_cuda or not t.is_pinned():
		raise _lib.NovicHipError("host_mapped_ptr: a pinned host tensor")
	out = ctypes.c_void_p()
	check(_lib.lib().novic_host_mapped_ptr(_vp(t.data_ptr()), ctypes.byref(out)), "novic_host_mapped_ptr")
	return int(out.value)


def step_done(active_word: torch.Tensor, flag_dev_ptr: int):
	"""One thread behind a decode step's selection kernel: *flag = 2 if *active_word != 0 else 1, straight into mapped host memory (novic_step_done): the decode loop's
	early-exit look without a device -> host copy or an event per step (embedding_decoder.py:819-820, :965-967)."""
	_dev(active_word)
	check(_lib.lib().novic_step_done(_ptr(active_word), _vp(flag_dev_ptr), _stream()), "novic_step_done")


def greedy_finalize(ids, pad, score, count, B, G, alpha):
	check(_lib.lib().novic_greedy_finalize(_ptr(ids), _tok_bytes(ids), _ptr(pad), _ptr(score), _ptr(count), B, G, ctypes.c_float(alpha), _stream()), "novic_greedy_finalize")


def beam_step(logits, ldl, V, B, H, G, step, ids_in, ids_out, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active, temperature, alpha, src_out=None,
              x_next=None, wtok=None, pos_row=None, origin_in=None, origin_out=None, npos=0):
	"""x_next (+ wtok, pos_row; origin_in / origin_out / npos): also write the next step's input rows and the K/V origin table of the new beams (novic_decode_embed and
	novic_kv_origin_update without launches of their own)."""
	ne = _next_embed(x_next, wtok, pos_row, origin_in, origin_out, npos)
	check(_lib.lib().novic_beam_step_next(_ptr(logits), ldl, V, B, H, G, step, _ptr(ids_in), _ptr(ids_out), _tok_bytes(ids_in), _ptr(pad_in), _ptr(pad_out), _ptr(score_in),
	                                      _ptr(score_out), _ptr(score_normed), _ptr(len_in), _ptr(len_out), _ptr(active), _ptr(src_out), ctypes.c_float(temperature),
	                                      ctypes.c_float(alpha), ctypes.byref(ne) if ne is not None else None, _stream()), "novic_beam_step")


def mask_ids(ids, pad):
	check(_lib.lib().novic_mask_ids(_ptr(ids), _tok_bytes(ids), _ptr(pad), ids.numel(), _stream()), "novic_mask_ids")


def vit_im2col(images: torch.Tensor, patches: torch.Tensor, patch: int, norm=None):
	"""images [B][3][R][R] -> rows of the bf16 patch matrix `patches` (which may be a row range of a larger buffer: coalesced image batches).  fp32 images are the
	transform's normalised output; uint8 images are its pixels before ToTensor / Normalize and need norm = (mean[3], std[3]): the kernel applies both steps per pixel."""
	_dev(images, patches)
	B, _, R, _ = images.shape
	assert patches.is_contiguous() and patches.shape[0] == B * (R // patch) ** 2
	if images.dtype == torch.uint8:
		if norm is None:
			raise ValueError("uint8 images need the transform's (mean, std)")
		pn = _lib.PixelNorm((ctypes.c_float * 3)(*[float(v) for v in norm[0]]), (ctypes.c_float * 3)(*[float(v) for v in norm[1]]))
		check(_lib.lib().novic_vit_im2col_u8(_ptr(images), _ptr(patches), B, R, patch, patches.shape[1], pn, _stream()), "novic_vit_im2col_u8")
		return
	assert images.dtype == torch.float32
	check(_lib.lib().novic_vit_im2col(_ptr(images), _ptr(patches), B, R, patch, patches.shape[1], _stream()), "novic_vit_im2col")


def vit_embed(patches, cls, pos, ln_w, ln_b, x, B, N, W, eps=1e-5):
	if x.dtype == torch.float16:  # (a half-precision residual stream: novic_vit_embed_f16)
		check(_lib.lib().novic_vit_embed_f16(_ptr(patches), _ptr(cls), _ptr(pos), _ptr(ln_w), _ptr(ln_b), _ptr(x), B, N, W, ctypes.c_float(eps), _stream()), "novic_vit_embed_f16")
		return
	assert x.dtype == torch.float32
	check(_lib.lib().novic_vit_embed(_ptr(patches), _ptr(cls), _ptr(pos), _ptr(ln_w), _ptr(ln_b), _ptr(x), B, N, W, ctypes.c_float(eps), _stream()), "novic_vit_embed")


def vit_attn_policy(policy: int = -1) -> int:
	"""0: streaming attention kernel only, 1: K/V-resident kernel where it fits (default); returns the previous policy (-1 only queries)."""
	return int(_lib.lib().novic_vit_attn_policy(int(policy)))


def vit_attn_fwd(qkv, o, B, N, H, D, scale: Optional[float] = None):
	"""scale: soft-max scale when it is not D ** -0.5 (heads zero-padded from their real width to D)."""
	if scale is not None:
		return clip_attn_fwd(qkv, o, B, N, H, D, causal=False, scale=scale)
	check(_lib.lib().novic_vit_attn_fwd(_ptr(qkv), _ptr(o), B, N, H, D, _stream()), "novic_vit_attn_fwd")


def rownorm_f32(x, y):
	check(_lib.lib().novic_rownorm_f32(_ptr(x), _ptr(y), x.shape[0], x.shape[1], _stream()), "novic_rownorm_f32")


def decode_embed(ids, G, col, wtok, pos_row, x, A, E, V):
	check(_lib.lib().novic_decode_embed(_ptr(ids), _tok_bytes(ids), G, col, _ptr(wtok), _ptr(pos_row), _ptr(x), A, E, V, _stream()), "novic_decode_embed")


def decode_attn(qkv_new, prefix_qkv, cache_k, cache_v, o, A, H, D, P, G, pos, beams, origin=None):
	check(_lib.lib().novic_decode_attn(_ptr(qkv_new), _ptr(prefix_qkv), _ptr(cache_k), _ptr(cache_v), _ptr(o), A, H, D, P, G, pos, beams, _ptr(origin), _stream()), "novic_decode_attn")


def kv_origin_update(src_idx, origin_in, origin_out, A, beams, G, npos):
	check(_lib.lib().novic_kv_origin_update(_ptr(src_idx), _ptr(origin_in), _ptr(origin_out), A, beams, G, npos, _stream()), "novic_kv_origin_update")


def kv_reorder(k_in, v_in, k_out, v_out, src_idx, layers, A, beams, G, E, npos):
	check(_lib.lib().novic_kv_reorder(_ptr(k_in), _ptr(v_in), _ptr(k_out), _ptr(v_out), _ptr(src_idx), layers, A, beams, G, E, npos, _stream()), "novic_kv_reorder")


def cache_gather(embeds, ids, tok, msk, wts, start, B, N, F, M_file, C_file, M, C, o_embed, o_target, o_mask, o_weight, weight_mode: int, staged_row0: int = -1):
	_dev(embeds, o_embed)
	tb = _tok_bytes(tok) if tok is not None else 8
	check(_lib.lib().novic_cache_gather(_ptr(embeds), _ptr(ids), _ptr(tok), tb, _ptr(msk), _ptr(wts), ctypes.c_int64(start), B, ctypes.c_int64(N), F, M_file, C_file, M, C,
	                                    _ptr(o_embed), _ptr(o_target), _ptr(o_mask), _ptr(o_weight), int(weight_mode), ctypes.c_int64(staged_row0), _stream()), "novic_cache_gather")


def cache_gather_group(embeds, ids, tok, msk, wts, starts, B_each, N, F, M_file, C_file, M, C, o_embed, o_target, o_mask, o_weight, weight_mode: int, staged_row0: int = -1):
	"""len(starts) batches of B_each rows in one launch, outputs batch after batch (novic_cache_gather_group)."""
	_dev(embeds, o_embed)
	tb = _tok_bytes(tok) if tok is not None else 8
	arr = (ctypes.c_int64 * len(starts))(*[int(v) for v in starts])
	check(_lib.lib().novic_cache_gather_group(_ptr(embeds), _ptr(ids), _ptr(tok), tb, _ptr(msk), _ptr(wts), arr, len(starts), int(B_each), ctypes.c_int64(N), F, M_file, C_file, M, C,
	                                          _ptr(o_embed), _ptr(o_target), _ptr(o_mask), _ptr(o_weight), int(weight_mode), ctypes.c_int64(staged_row0), _stream()), "novic_cache_gather_group")


def beam_step_guided(logits, ldl, V, B, H, G, step, ids_in, ids_out, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active, src_out, node_in, node_out, trie,
                     logprior, prior_scale, renorm, temperature, alpha):
	check(_lib.lib().novic_beam_step_guided(_ptr(logits), ldl, V, B, H, G, step, _ptr(ids_in), _ptr(ids_out), _tok_bytes(ids_in), _ptr(pad_in), _ptr(pad_out), _ptr(score_in),
	                                        _ptr(score_out), _ptr(score_normed), _ptr(len_in), _ptr(len_out), _ptr(active), _ptr(src_out), _ptr(node_in), _ptr(node_out),
	                                        _ptr(trie.start), _ptr(trie.tok), _ptr(trie.next), _ptr(logprior), ctypes.c_float(prior_scale), int(renorm), ctypes.c_float(temperature),
	                                        ctypes.c_float(alpha), _stream()), "novic_beam_step_guided")


def greedy_step_guided(logits, ldl, V, B, G, step, ids, pad, alive, score, nll, count, active, step_logits, node, trie, renorm, temperature, smoothing):
	check(_lib.lib().novic_greedy_step_guided(_ptr(logits), ldl, V, B, G, step, _ptr(ids), _tok_bytes(ids), _ptr(pad), _ptr(alive), _ptr(score), _ptr(nll), _ptr(count),
	                                          _ptr(active), _ptr(step_logits), _ptr(node), _ptr(trie.start), _ptr(trie.tok), _ptr(trie.next), int(renorm),
	                                          ctypes.c_float(temperature), ctypes.c_float(smoothing), _stream()), "novic_greedy_step_guided")


def gemm_tile_policy(policy: int = -1) -> int:
	"""0: 128^2-tile kernel only, 1: large problems on the 256^2-tile kernel (default); returns the previous policy (-1 only queries)."""
	return int(_lib.lib().novic_gemm_tile_policy(int(policy)))


def gemm_last_tile() -> int:
	"""128 or 256: which GEMM kernel the most recent gemm() call launched."""
	return int(_lib.lib().novic_gemm_last_tile())


def gemm256_pipeline(on: int = -1) -> int:
	"""1: the 256 x 256 tile runs the 8-phase K loop (default), 0: one barrier per K-tile; returns the previous setting (-1 only queries)."""
	return int(_lib.lib().novic_gemm256_pipeline(int(on)))


def gemm256_plan(M: int, N: int, K: int, *, kind=EPI_STORE_BF16, act=ACT_NONE, bias: bool = False, row_limit: bool = False, split_tail: bool = False,
                 scratch_bytes: int = 256 << 20) -> dict:
	"""What gemm() would choose for a K-contiguous [M x N x K] problem once it reaches the 256-wide kernels (novic_gemm256_plan: the decision alone -- no launch, no
	GPU): tile width (0 = left to the 128 x 128 kernel), workgroups, K-split of the tail tiles (parts, -1 = planned on the device from the row count, 0 = none), tail tiles."""
	ep = Epilogue()
	ep.struct_bytes = ctypes.sizeof(Epilogue)
	ep.kind, ep.act = kind, act
	ep.c, ep.bias = 0x100000, (0x200000 if bias else 0)  # (only null-ness and alignment are looked at)
	ep.resid = 0x300000 if kind in (EPI_RESID_F32, EPI_RESID_F16) else 0
	ep.ldc = ep.ldr = N
	ep.alpha = 1.0
	ep.row_limit = 0x400000 if row_limit else 0
	ep.max_workgroups = getattr(_tls, "cus", 0)
	if split_tail:
		ep.splitk_ws, ep.splitk_ws_bytes = 0x500000, int(scratch_bytes)
	out = (ctypes.c_int * 4)()
	check(_lib.lib().novic_gemm256_plan(int(M), int(N), int(K), ctypes.byref(ep), out), "novic_gemm256_plan")
	return dict(tile=int(out[0]), workgroups=int(out[1]), tail_parts=int(out[2]), tail_tiles=int(out[3]))


def persistent_cus(n: int = -1) -> int:
	"""The library's process-wide DEFAULT workgroup budget per persistent 256-wide GEMM grid (multiple of 8 in 8..256, 256 = every CU, start value $NOVIC_PERSISTENT_CUS); a
	negative n only queries.  Returns the previous value.  Product code uses `cu_budget` (per call) instead of changing this."""
	return int(_lib.lib().novic_persistent_cus(int(n)))


def gemm_tile_counts(reset: bool = False) -> dict:
	"""gemm() launches per kernel since the last reset (novic_gemm_tile_counts)."""
	buf = (ctypes.c_ulonglong * 7)()
	check(_lib.lib().novic_gemm_tile_counts(buf, int(reset)), "novic_gemm_tile_counts")
	return dict(zip(("t128", "skinny", "t256", "t192", "ksplit_tail", "ksplit_tail_device", "unused"), (int(v) for v in buf)))


def decode_fused_supported(E: int, Kf: int) -> bool:
	return bool(_lib.lib().novic_decode_fused_supported(int(E), int(Kf)))


def decode_ln_gemm(x: torch.Tensor, gamma: torch.Tensor, w: torch.Tensor, y: torch.Tensor, M: int, N: int, E: int, gelu: bool = False, eps: float = 1e-5):
	_dev(x, gamma, w, y)
	check(_lib.lib().novic_decode_ln_gemm(_ptr(x), _ptr(gamma), _ptr(w), _ptr(y), M, N, E, y.stride(0), int(gelu), ctypes.c_float(eps), _stream()), "novic_decode_ln_gemm")


def decode_gemm_resid(a: torch.Tensor, w: torch.Tensor, resid: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int):
	_dev(a, w, resid, out)
	check(_lib.lib().novic_decode_gemm_resid(_ptr(a), _ptr(w), _ptr(resid), _ptr(out), M, N, K, _stream()), "novic_decode_gemm_resid")


def decode_ffn_supported(E: int, Kf: int) -> bool:
	return bool(_lib.lib().novic_decode_ffn_supported(int(E), int(Kf)))


def decode_ffn(x: torch.Tensor, gamma: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, out: torch.Tensor, M: int, E: int, Kf: int, eps: float = 1e-5):
	"""out = x + GELU(LayerNorm(x) w1^T) w2^T in one launch (novic_decode_ffn); out must be another buffer than x."""
	_dev(x, gamma, w1, w2, out)
	check(_lib.lib().novic_decode_ffn(_ptr(x), _ptr(gamma), _ptr(w1), _ptr(w2), _ptr(out), M, E, Kf, ctypes.c_float(eps), _stream()), "novic_decode_ffn")


def decode_gemm(a: torch.Tensor, w: torch.Tensor, y: torch.Tensor, M: int, N: int, K: int, gelu: bool = False):
	_dev(a, w, y)
	check(_lib.lib().novic_decode_gemm(_ptr(a), _ptr(w), _ptr(y), M, N, K, y.stride(0), int(gelu), _stream()), "novic_decode_gemm")


def score_targets(logits: torch.Tensor, ldl: int, V: int, targets: torch.Tensor, pad: torch.Tensor, node: Optional[torch.Tensor], trie, out: torch.Tensor, w0: int, B: int, Hc: int,
                  T: int, temperature: float):
	_dev(logits, targets, pad, out)
	check(_lib.lib().novic_score_targets(_ptr(logits), ldl, V, _ptr(targets), _tok_bytes(targets), _ptr(pad), _ptr(node), _ptr(trie.start) if node is not None else None,
	                                     _ptr(trie.tok) if node is not None else None, _ptr(out), out.stride(0), w0, B, Hc, T, ctypes.c_float(temperature), _stream()),
	      "novic_score_targets")


def topk_rows(scores: torch.Tensor, K: int, out_val: torch.Tensor, out_idx: torch.Tensor, adjust: Optional[torch.Tensor] = None, adjust_scale: float = 0.0,
              scale: Optional[torch.Tensor] = None):
	_dev(scores, out_val, out_idx)
	B, W = scores.shape
	check(_lib.lib().novic_topk_rows(_ptr(scores), B, W, scores.stride(0), _ptr(adjust), ctypes.c_float(adjust_scale), _ptr(scale), K, _ptr(out_val), _ptr(out_idx), _stream()),
	      "novic_topk_rows")


def clip_attn_fwd(qkv: torch.Tensor, o: torch.Tensor, B: int, N: int, H: int, D: int, causal: bool, scale: Optional[float] = None):
	_dev(qkv, o)
	if scale is not None:
		check(_lib.lib().novic_clip_attn_fwd_scaled(_ptr(qkv), _ptr(o), B, N, H, D, int(causal), ctypes.c_float(scale), _stream()), "novic_clip_attn_fwd_scaled")
		return
	check(_lib.lib().novic_clip_attn_fwd(_ptr(qkv), _ptr(o), B, N, H, D, int(causal), _stream()), "novic_clip_attn_fwd")


def text_embed(ids: torch.Tensor, tok_emb: torch.Tensor, pos: torch.Tensor, x: torch.Tensor, B: int, S: int, W: int):
	_dev(ids, tok_emb, pos, x)
	if x.dtype == torch.float16:  # (a half-precision residual stream: novic_text_embed_f16)
		check(_lib.lib().novic_text_embed_f16(_ptr(ids), _tok_bytes(ids), _ptr(tok_emb), _ptr(pos), _ptr(x), B, S, W, tok_emb.shape[0], _stream()), "novic_text_embed_f16")
		return
	assert x.dtype == torch.float32
	check(_lib.lib().novic_text_embed(_ptr(ids), _tok_bytes(ids), _ptr(tok_emb), _ptr(pos), _ptr(x), B, S, W, tok_emb.shape[0], _stream()), "novic_text_embed")


def text_pool(ids: torch.Tensor, x: torch.Tensor, out: torch.Tensor, B: int, S: int, W: int, eot_id: int = -1):
	_dev(ids, x, out)
	if x.dtype == torch.float16:
		check(_lib.lib().novic_text_pool_f16(_ptr(ids), _tok_bytes(ids), _ptr(x), _ptr(out), B, S, W, ctypes.c_longlong(eot_id), _stream()), "novic_text_pool_f16")
		return
	assert x.dtype == torch.float32
	check(_lib.lib().novic_text_pool(_ptr(ids), _tok_bytes(ids), _ptr(x), _ptr(out), B, S, W, ctypes.c_longlong(eot_id), _stream()), "novic_text_pool")


def guided_correct(logits: torch.Tensor, ldl: int, targets: torch.Tensor, tok_ld: int, out_pad: Optional[torch.Tensor], trie, correct: torch.Tensor, A: int, T: int):
	_dev(logits, targets, correct)
	check(_lib.lib().novic_guided_correct(_ptr(logits), ldl, _ptr(targets), _tok_bytes(targets), tok_ld, _ptr(out_pad), _ptr(trie.start), _ptr(trie.tok), _ptr(trie.next),
	                                      _ptr(correct), A, T, _stream()), "novic_guided_correct")


def beam_step_guided_vocab(logits, ldl, V, B, H, G, step, ids_in, ids_out, pad_in, pad_out, score_in, score_out, score_normed, len_in, len_out, active, src_out, node_in, node_out, trie,
                           vnode_in, vnode_out, vtrie, vlogprior, prior_scale, renorm, temperature, alpha):
	check(_lib.lib().novic_beam_step_guided_vocab(_ptr(logits), ldl, V, B, H, G, step, _ptr(ids_in), _ptr(ids_out), _tok_bytes(ids_in), _ptr(pad_in), _ptr(pad_out), _ptr(score_in),
	                                              _ptr(score_out), _ptr(score_normed), _ptr(len_in), _ptr(len_out), _ptr(active), _ptr(src_out), _ptr(node_in), _ptr(node_out),
	                                              _ptr(trie.start), _ptr(trie.tok), _ptr(trie.next), _ptr(vnode_in), _ptr(vnode_out), _ptr(vtrie.start), _ptr(vtrie.tok), _ptr(vtrie.next),
	                                              _ptr(vlogprior), ctypes.c_float(prior_scale), int(renorm), ctypes.c_float(temperature), ctypes.c_float(alpha), _stream()),
	      "novic_beam_step_guided_vocab")


def ffn_fused_supported(E: int, Kf: int, M: int = 0) -> bool:
	"""The fused feed-forward launches exist for hidden 512 / feed-forward 128 and address rows through 32-bit buffer offsets: M rows of 2 KiB must stay below 4 GiB."""
	return bool(_lib.lib().novic_ffn_fused_supported(int(E), int(Kf))) and M * E * 4 < 0xFFFFFFF0


def ffn_fwd(xmid: torch.Tensor, gamma2: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, x_out: torch.Tensor, M: int, E: int, Kf: int, *, gamma_next: Optional[torch.Tensor] = None,
            ln_next: Optional[torch.Tensor] = None, ln2: Optional[torch.Tensor] = None, hpre: Optional[torch.Tensor] = None, hact: Optional[torch.Tensor] = None, eps: float = 1e-5,
            dropout: Dropout = NO_DROPOUT, site_gelu: int = 0, site_out: int = 0, row_limit: Optional[torch.Tensor] = None):
	"""LayerNorm + linear1 + GELU + dropout + linear2 + dropout + residual (+ the next layer's LayerNorm) in one launch (novic_ffn_fwd, csrc/ffn.hip)."""
	_dev(xmid, gamma2, w1, w2, x_out)
	assert w1.is_contiguous() and w2.is_contiguous() and xmid.is_contiguous() and x_out.is_contiguous()
	check(_lib.lib().novic_ffn_fwd(_ptr(xmid), _ptr(gamma2), _ptr(w1), _ptr(w2), _ptr(gamma_next), _ptr(x_out), _ptr(ln2), _ptr(hpre), _ptr(hact), _ptr(ln_next), M, E, Kf,
	                               ctypes.c_float(eps), ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(site_gelu), ctypes.c_uint32(site_out), _ptr(row_limit), _stream()),
	      "novic_ffn_fwd")


def ffn_bwd(gb: torch.Tensor, hpre: torch.Tensor, xmid: torch.Tensor, dx_in: torch.Tensor, gamma2: torch.Tensor, w2t: torch.Tensor, w1t: torch.Tensor, dh: torch.Tensor,
            dx_out: torch.Tensor, g_out: torch.Tensor, dgamma2: torch.Tensor, M: int, E: int, Kf: int, *, eps: float = 1e-5, dropout: Dropout = NO_DROPOUT, site_gelu: int = 0,
            site_g: int = 0, row_limit: Optional[torch.Tensor] = None):
	"""linear2 input gradient + GELU' + linear1 input gradient + LayerNorm backward in one launch (novic_ffn_bwd, csrc/ffn.hip)."""
	_dev(gb, hpre, xmid, dx_in, dh, dx_out, g_out, dgamma2)
	assert w2t.is_contiguous() and w1t.is_contiguous() and gb.is_contiguous() and hpre.is_contiguous()
	check(_lib.lib().novic_ffn_bwd(_ptr(gb), _ptr(hpre), _ptr(xmid), _ptr(dx_in), _ptr(gamma2), _ptr(w2t), _ptr(w1t), _ptr(dh), _ptr(dx_out), _ptr(g_out), _ptr(dgamma2), M, E, Kf,
	                               ctypes.c_float(eps), ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(site_gelu), ctypes.c_uint32(site_g), _ptr(row_limit), _stream()),
	      "novic_ffn_bwd")


def ffn_bwd_ln(pre_dln: torch.Tensor, pre_x: torch.Tensor, pre_gamma: torch.Tensor, pre_dgamma: torch.Tensor, gb_out: torch.Tensor, hpre: torch.Tensor, xmid: torch.Tensor,
               dx_in: Optional[torch.Tensor], gamma2: torch.Tensor, w2t: torch.Tensor, w1t: torch.Tensor, dh: torch.Tensor, dx_out: torch.Tensor, g_out: torch.Tensor, dgamma2: torch.Tensor,
               M: int, E: int, Kf: int, *, eps: float = 1e-5, dropout: Dropout = NO_DROPOUT, site_pre: int = 0, site_gelu: int = 0, site_g: int = 0,
               row_limit: Optional[torch.Tensor] = None, pre_row_map: Optional[torch.Tensor] = None):
	"""The LayerNorm backward of the layer above (its norm1; or the final norm, with pre_row_map picking the compacted output rows and dx_in None) as the prologue of
	ffn_bwd: one launch (novic_ffn_bwd_ln, csrc/ffn.hip)."""
	_dev(pre_dln, pre_x, pre_dgamma, gb_out, hpre, xmid, dh, dx_out, g_out, dgamma2)
	assert pre_row_map is None or (pre_row_map.dtype == torch.int32 and pre_row_map.is_cuda)
	assert w2t.is_contiguous() and w1t.is_contiguous() and pre_dln.is_contiguous() and pre_x.is_contiguous() and gb_out.is_contiguous() and hpre.is_contiguous()
	check(_lib.lib().novic_ffn_bwd_ln(_ptr(pre_dln), _ptr(pre_row_map), _ptr(pre_x), _ptr(pre_gamma), _ptr(pre_dgamma), _ptr(gb_out), ctypes.c_uint32(site_pre), _ptr(hpre), _ptr(xmid), _ptr(dx_in),
	                                  _ptr(gamma2), _ptr(w2t), _ptr(w1t), _ptr(dh), _ptr(dx_out), _ptr(g_out), _ptr(dgamma2), M, E, Kf, ctypes.c_float(eps),
	                                  ctypes.c_float(dropout.p), _u64(dropout.seed), ctypes.c_uint32(site_gelu), ctypes.c_uint32(site_g), _ptr(row_limit),
	                                  _stream()), "novic_ffn_bwd_ln")


def beam_step_policy(generic: int = -1) -> int:
	"""1: force the workgroup-per-sample beam step kernel, 0: one wave per beam row where V <= 8192 (default); returns the previous setting (-1 only queries)."""
	return int(_lib.lib().novic_beam_step_policy(int(generic)))


def skinny_wide_policy(wide: int = -1) -> int:
	"""0: four 128-column blocks (default), 1: two 256-column blocks for the [M x 512 x 512] bf16-store GEMM of the streaming kernel; returns the previous setting."""
	return int(_lib.lib().novic_skinny_wide_policy(int(wide)))
