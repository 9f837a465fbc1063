"""What the three native towers (clip_vit.NativeViT, clip_text.NativeTextTower, siglip.NativeSigLIPViT) share around their launch sequences: a workspace per batch shape
and the hipGraph replay of that shape's launches.  No reference counterpart (the reference calls open_clip / clip / transformers modules: embedders.py:589-594, :759-764).

One SLOT per (input shape, dtype, normalize, device, workgroup budget): the slot owns every buffer a forward of that shape touches -- activations, the K-split scratch of its
GEMMs (one per lane: `ops.splitk_scratch`), the captured graph, its static input / output.  A captured hipGraph holds the ADDRESSES of its slot's buffers, so graph and
buffers live and die together: slots are evicted least-recently-used, beyond `max_shapes` shapes or `max_ws_bytes` bytes, after a device synchronisation (a replay may
still be running out of them).  Round 3 kept one workspace per buffer NAME in two towers' first version (a ragged last batch replaced the buffers a captured graph pointed
to: a GPU fault) and the SigLIP trunk never replayed a graph at all; this module is the one implementation all three now use
(tests/test_gpu_vit.py::test_tower_graphs_survive_other_batch_shapes runs over all of them).
"""
from __future__ import annotations

import contextlib
from collections import OrderedDict

import torch

from . import ops


class _Slot:
	__slots__ = ("ws", "graph", "static_in", "out", "calls")

	def __init__(self):
		self.ws: dict = {}
		self.graph = None
		self.static_in = None
		self.out = None
		self.calls = 0

	def nbytes(self) -> int:
		return sum(t.numel() * t.element_size() for t in self.ws.values())

	def __del__(self):  # a captured graph is never destroyed while a capture is open on any thread (ops.retire_graphs)
		try:
			g, self.graph = self.graph, None
			if g is not None:
				ops.retire_graphs([g])
		except Exception:  # noqa: BLE001 -- interpreter shutdown: modules may be gone
			pass


class TowerRuntime:
	# From the second call with a given batch shape on, the launch sequence -- static for a shape -- is replayed from a captured hipGraph: one forward is ~90 launches of
	# 15-70 us issued through ctypes from Python at ~40 us per call (ViT-B/32 at batch 256: 2.46 ms of kernels, 3.85 ms per eager forward, the GPU idle a third of the time).
	use_graphs = True
	max_shapes = 3            # batch shapes kept (workspace + graph); a caller with ragged batches cycles through full / last-batch shapes: two
	max_ws_bytes = 48 << 30   # ... and at most this much workspace over all of them (ViT-H/14-378 at batch 256: ~7 GB per shape)

	def _rt_slots(self) -> "OrderedDict":
		d = self.__dict__.get("_slots")
		if d is None:
			d = self.__dict__["_slots"] = OrderedDict()
		return d

	def _rt_reset(self):
		"""The weights changed: every captured graph reads the old bf16 shadow, and some workspace buffers hold constants derived from it."""
		slots = self.__dict__.get("_slots")
		if slots:
			torch.cuda.synchronize()
			slots.clear()

	def _rt_slot(self, key, device) -> _Slot:
		slots = self._rt_slots()
		s = slots.get(key)
		if s is None:
			s = slots[key] = _Slot()
		else:
			slots.move_to_end(key)
		while len(slots) > 1 and (len(slots) > int(self.max_shapes) or sum(v.nbytes() for v in slots.values()) > int(self.max_ws_bytes)):
			torch.cuda.synchronize(device)  # (a replay may still be running out of the buffers about to be freed)
			slots.popitem(last=False)
		return s

	@contextlib.contextmanager
	def _rt_use(self, slot: _Slot):
		prev = self.__dict__.get("_cur_slot")
		self.__dict__["_cur_slot"] = slot
		try:
			yield slot
		finally:
			self.__dict__["_cur_slot"] = prev

	def _buf2(self, name, shape, dtype, device):
		"""(buffer `name` of the current slot, whether it was allocated by this call -- the caller then fills in whatever constant it must hold)."""
		slot = self.__dict__.get("_cur_slot")
		if slot is None:
			raise RuntimeError("tower workspace requested outside a forward")
		t = slot.ws.get(name)
		fresh = t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype or t.device != device
		if fresh:
			if t is not None and slot.graph is not None:
				raise RuntimeError(f"tower workspace '{name}' changed shape inside one slot: {tuple(t.shape)} -> {tuple(shape)}")
			t = slot.ws[name] = torch.empty(tuple(shape), dtype=dtype, device=device)
		return t, fresh

	def _buf(self, name, shape, dtype, device):
		return self._buf2(name, shape, dtype, device)[0]

	def _lane_scratch(self, lane: int, device):
		"""K-split scratch of this slot's lane (64 MiB): owned by the slot, so a captured graph's scratch is freed with the graph and never shared with another stream's."""
		return ops.splitk_scratch(self._buf(f"L{lane}:splitk", (16 * 1024 * 1024,), torch.float32, device))

	def _rt_forward(self, x: torch.Tensor, normalize: bool, eager, capture_tail=None, before_replay=None, static_input: bool = False, variant=()) -> torch.Tensor:
		"""eager(x) -> out: the whole launch sequence.  capture_tail(x) -> out: the part of it that a graph may hold (default: all of eager); before_replay(x): launches that
		read the CALLER's tensor and run in front of every replay (im2col into the slot's patch buffer).  static_input: the graph reads a slot-owned copy of x.
		The launch sequence may fork onto other streams and join again (lanes): a capture records that as branches of the graph."""
		# x: the input batch, or a LIST of batches that one forward runs as their concatenation (coalesced image batches: every callback receives the list; the slot is
		# keyed by the shapes in order, the dtype of the first)
		many = isinstance(x, (list, tuple))
		assert not (many and static_input)
		first = x[0] if many else x
		dev = first.device
		shape = tuple(tuple(t.shape) for t in x) if many else tuple(x.shape)
		key = (shape, first.dtype, bool(normalize), dev, ops.current_cu_budget(), bool(getattr(self, "half_stream", False))) + tuple(variant)  # (the grid sizes are baked into a capture; the stream's type selects other launches and buffers; variant: lanes, ...)
		slot = self._rt_slot(key, dev)
		with self._rt_use(slot):
			slot.calls += 1
			if not self.use_graphs or slot.calls == 1:  # first call with this shape: eager (it also allocates the workspace the capture will reuse)
				return eager(x)
			if slot.graph is None:
				# (outside inference mode: the static buffers are updated in place by later calls from either mode, and torch registers its generator state with the capture --
				# state tensors created by a capture INSIDE inference mode make every later capture outside it fail)
				with torch.inference_mode(False):
					src = x
					if static_input:
						slot.static_in = torch.empty_like(x)
						slot.static_in.copy_(x)
						src = slot.static_in
					cur = torch.cuda.current_stream(dev)
					side = ops.capture_stream(dev)
					side.wait_stream(cur)
					with torch.cuda.stream(side):
						g = torch.cuda.CUDAGraph()
						with ops.graph_capture(g, side):
							out = (capture_tail or eager)(src)
					cur.wait_stream(side)
				slot.graph, slot.out = g, out
			if static_input:
				slot.static_in.copy_(x)
			if before_replay is not None:
				before_replay(x)
			slot.graph.replay()
			return slot.out.clone()
