"""Embedder surface: target-token configuration + (de)tokenisation + the image-embedding entry point.

Mirrors reference embedders.py: ``TargetConfig`` (:42-65), ``Embedder`` base API (:68-435: ``create_target_config`` :169-254,
``configure_target`` :256, ``tokenize_target`` :331-385, ``detokenize_target`` :387-406, ``inference_mode`` :295-306,
``inference_image`` :432).  The reference's three back-ends wrap third-party CLIP packages that fetch weights by name; none of
them is reachable here, so this module provides

* ``Embedder``: the back-end-independent logic (compact-id vocabulary, target tokenisation, masks) over an abstract tokenizer;
* ``LocalVocabEmbedder``: a self-contained word-piece-free tokenizer over an explicit token list (plumbing / tests / synthetic
  checkpoints), with the same special-token conventions as the CLIP tokenizers (start, end, pad = 0 after compaction);
* the image path (``inference_image``) is served by ``novic_amd.clip_vit.NativeViT`` -- hand-written HIP kernels -- when an
  image tower is attached via ``attach_image_tower``;
* ``TransformersEmbedder`` (reference :767-907) for a LOCAL Hugging Face CLIP directory, spec ``'transformers:/path/to/dir'``: transformers'
  tokenizer on the host, the directory's weights in the native image and text towers;
* ``OpenCLIPEmbedder`` / ``OpenAIEmbedder`` (reference :438-764, ``novic_amd/local_clip.py``): ``'openclip:ORG/NAME'`` / ``'openai:ViT-B/32'`` resolved against local
  storage only ($NOVIC_MODEL_ROOT, the Hugging Face hub cache, clip's download directory) -- never the network.
"""
from __future__ import annotations

import contextlib
import dataclasses
import hashlib
import itertools
import json
import os
from typing import Any, Optional, Sequence, Union

import torch


@dataclasses.dataclass(frozen=True)
class TargetConfig:
	vocab_size: int
	token_dtype: torch.dtype
	mask_dtype: torch.dtype
	start_token_id: Optional[int]
	end_token_id: Optional[int]
	pad_token_id: int
	compact_ids: bool
	compact_map: Optional[torch.Tensor]
	compact_unmap: Optional[torch.Tensor]
	fixed_token_length: bool
	token_length: int
	use_masks: bool

	def _scalars(self):
		return (self.vocab_size, self.token_dtype, self.mask_dtype, self.start_token_id, self.end_token_id, self.pad_token_id, self.compact_ids,
		        self.fixed_token_length, self.token_length, self.use_masks)

	def __eq__(self, other):
		if other.__class__ is not self.__class__:
			return NotImplemented

		def same(a, b):
			return a is b or (a is not None and b is not None and a.dtype == b.dtype and torch.equal(a, b))
		return self._scalars() == other._scalars() and same(self.compact_map, other.compact_map) and same(self.compact_unmap, other.compact_unmap)

	__hash__ = None


class Embedder:
	"""Back-end independent part of the reference's Embedder.  Subclasses provide tokenize()/detokenize()."""

	@staticmethod
	def create(spec: str, amp: bool = True, amp_bf16: bool = False, tokenizer_batch_size: int = 1024, inference_batch_size: int = 256, image_batch_size: int = 128,
	           load_model: bool = True, compile_model: bool = False, use_optimum: bool = False, device: Union[int, str, torch.device] = "cuda", check: bool = False) -> "Embedder":
		if ":" not in spec:
			raise ValueError(f"Embedder spec must be of the format 'TYPE:NAME': {spec}")
		kind, name = spec.split(":", maxsplit=1)
		if kind == "local":  # 'local:/path/to/embedder.json' -- vocabulary (+ optional ViT weights) from local files only
			return LocalVocabEmbedder.from_file(name, amp=amp, amp_bf16=amp_bf16, tokenizer_batch_size=tokenizer_batch_size, inference_batch_size=inference_batch_size,
			                                    image_batch_size=image_batch_size, load_model=load_model, device=device, check=check)
		if kind == "openclip":  # reference :597-764 -- the hub repository's files from local storage ($NOVIC_MODEL_ROOT/ORG/NAME, the HF hub cache, or a directory path)
			from .local_clip import OpenCLIPEmbedder
			return OpenCLIPEmbedder(name, amp=amp, amp_bf16=amp_bf16, tokenizer_batch_size=tokenizer_batch_size, inference_batch_size=inference_batch_size,
			                        image_batch_size=image_batch_size, load_model=load_model, compile_model=compile_model, device=device, check=check)
		if kind == "openai":  # reference :438-594 -- the .pt file clip.load would download + CLIP's BPE vocabulary, from local storage
			from .local_clip import OpenAIEmbedder
			return OpenAIEmbedder(name, tokenizer_batch_size=tokenizer_batch_size, inference_batch_size=inference_batch_size, image_batch_size=image_batch_size,
			                      load_model=load_model, compile_model=compile_model, device=device, check=check)
		if kind == "transformers":  # a LOCAL Hugging Face CLIP directory: tokenizer on the host, both towers on the native kernels
			if not os.path.isdir(name):
				from .local_clip import resolve_model_dir
				name = resolve_model_dir(name, marker="config.json")
			return TransformersEmbedder(name, amp=amp, amp_bf16=amp_bf16, tokenizer_batch_size=tokenizer_batch_size, inference_batch_size=inference_batch_size,
			                            image_batch_size=image_batch_size, load_model=load_model, compile_model=compile_model, use_optimum=use_optimum, device=device, check=check)
		raise ValueError(f"Unsupported embedder type: {kind}")

	def __init__(self, configuration: dict[str, Any], context_length: int, vocab_size: int, cased_tokens: bool, start_token_id: Optional[int], end_token_id: int,
	             pad_token_id: int, token_dtype: torch.dtype, embed_dtype: torch.dtype, embed_dim: int, amp_mode: Union[bool, torch.dtype] = True,
	             manual_amp_dtype: Optional[torch.dtype] = None, tokenizer_batch_size: int = 1024, inference_batch_size: int = 256, image_batch_size: int = 128,
	             load_model: bool = True, compile_model: bool = False, device: Union[int, str, torch.device] = "cuda", check: bool = False):
		self.context_length, self.vocab_size, self.cased_tokens = context_length, vocab_size, cased_tokens
		self.start_token_id, self.end_token_id, self.pad_token_id = start_token_id, end_token_id, pad_token_id
		assert (isinstance(start_token_id, int) or start_token_id is None) and isinstance(end_token_id, int) and isinstance(pad_token_id, int)
		self.device = device if isinstance(device, torch.device) else torch.device(device)
		self.amp_mode = amp_mode
		# The native image tower computes in bf16 MFMA with fp32 accumulation; amp_dtype records that for parity with the reference's field
		self.amp_dtype = (amp_mode if isinstance(amp_mode, torch.dtype) else torch.bfloat16) if (amp_mode and self.device.type != "cpu") else None
		self.manual_amp_dtype = manual_amp_dtype
		self.token_dtype, self.embed_dtype, self.embed_dim = token_dtype, embed_dtype, embed_dim
		self.tokenizer_batch_size, self.inference_batch_size, self.image_batch_size = tokenizer_batch_size, inference_batch_size, image_batch_size
		self.configuration = dict(configuration)
		self.configuration["class"] = self.__class__.__qualname__
		self.configuration["device_type"] = self.device.type
		self.configuration["amp_dtype"] = format(self.amp_dtype)
		self.target_config: Optional[TargetConfig] = None
		self.target_vocab: Optional[tuple[str, ...]] = None
		self.target_configuration: Optional[dict[str, Any]] = None
		self.check = check
		self.compile_model = compile_model
		self.image_tower = None
		self._inference = False
		if load_model:
			self.load_model()

	# ---- model lifecycle (reference :281-318) ----
	def load_model(self) -> bool:
		return False

	def unload_model(self) -> bool:
		return False

	def is_model_loaded(self) -> bool:
		return self.image_tower is not None

	@contextlib.contextmanager
	def inference_model(self, release: bool = True):
		loaded = self.load_model()
		try:
			yield
		finally:
			if loaded and release:
				self.unload_model()

	@contextlib.contextmanager
	def inference_mode(self):
		prev, self._inference = self._inference, True
		try:
			with torch.inference_mode():
				yield
		finally:
			self._inference = prev

	def get_configuration(self, main_config: bool = True, target_config: bool = False, target_exclude: Optional[set] = None, target_override: Optional[dict] = None) -> dict[str, Any]:
		"""reference :262-273"""
		cfg = dict(self.configuration) if main_config else {}
		if target_config:
			if self.target_config is None or self.target_configuration is None:
				raise ValueError("Cannot get configuration including target config because there is none yet")
			cfg["target_config"] = {k: v for k, v in self.target_configuration.items() if not target_exclude or k not in target_exclude}
			if target_override is not None:
				cfg["target_config"].update(target_override)
		return cfg

	def get_configuration_hash(self, main_config: bool = True, target_config: bool = False, target_exclude: Optional[set] = None, target_override: Optional[dict] = None,
	                           hexdigest: bool = True, algorithm: str = "sha256"):
		"""Same bytes as the reference's hash (:275-278: compact sorted JSON) so caches written by either side validate against an identical configuration."""
		h = hashlib.new(name=algorithm, data=json.dumps(self.get_configuration(main_config, target_config, target_exclude, target_override), separators=(",", ":"), sort_keys=True).encode())
		return h.hexdigest() if hexdigest else h.digest()

	# ---- tokenizer interface to be provided by subclasses ----
	def tokenize(self, text: Union[str, Sequence[str]], max_tokens: Optional[int] = None, output_dict: bool = False):
		"""-> B x L ids (start? content... end, then pad) and, with output_dict, {'input_ids', 'attention_mask'} (mask 1 up to and including end)."""
		raise NotImplementedError

	def detokenize(self, token_ids: torch.Tensor) -> Union[str, list[str]]:
		raise NotImplementedError

	def get_tokenize_details(self, text: Union[str, Sequence[str]], max_tokens: Optional[int] = None, token_id_set: bool = False):
		out = self.tokenize(text, max_tokens=max_tokens, output_dict=True)
		ids, att = out["input_ids"], out["attention_mask"]
		lens = att.sum(dim=1)
		idx = int(lens.argmax())
		texts = (text,) if isinstance(text, str) else text
		ids_set = set(ids[att.bool()].tolist()) | {self.pad_token_id} if token_id_set else None
		return int(lens[idx]), texts[idx], ids_set

	# ---- target configuration (reference :169-254) ----
	def create_target_config(self, targets: Sequence[str], *, with_start_token: bool, with_end_token: bool, compact_ids: bool, fixed_token_length: bool,
	                         auto_fixed_token_length: bool, use_masks: bool) -> TargetConfig:
		used: set[int] = set()
		longest = 0
		it = iter(targets)
		while chunk := tuple(itertools.islice(it, self.tokenizer_batch_size)):
			n, _, ids = self.get_tokenize_details(chunk, token_id_set=compact_ids)
			if ids:
				used.update(ids)
			longest = max(longest, n)
		if compact_ids:
			used.remove(self.end_token_id)
		if not with_end_token:
			longest -= 1
		used.discard(self.pad_token_id)
		if self.start_token_id is None:
			if with_start_token:
				longest += 1
		else:
			if compact_ids:
				used.remove(self.start_token_id)
			if not with_start_token:
				longest -= 1
		if compact_ids:
			pad_id, end_id = 0, (0 if with_end_token else None)
			table = [self.pad_token_id]
			start_id = None
			if with_start_token:
				start_id = 1
				table.append(self.start_token_id if self.start_token_id is not None else -1)
			n_special = len(table)
			table.extend(sorted(used))
			vocab = len(table)
			unmap = torch.tensor(table, dtype=self.token_dtype)
			cmap = torch.full((self.vocab_size,), -1, dtype=self.token_dtype)
			cmap[unmap[n_special:].long()] = torch.arange(n_special, vocab, dtype=self.token_dtype)
			cmap[self.pad_token_id] = 0
			cmap[self.end_token_id] = 0
			if self.start_token_id is not None and with_start_token:
				cmap[self.start_token_id] = 1
		else:
			vocab = self.vocab_size
			start_id = self.start_token_id if with_start_token else None
			end_id = self.end_token_id if with_end_token else None
			pad_id, cmap, unmap = self.pad_token_id, None, None
		token_length = longest if (not fixed_token_length or auto_fixed_token_length) else self.context_length
		return TargetConfig(vocab_size=vocab, token_dtype=self.token_dtype, mask_dtype=torch.bool, start_token_id=start_id, end_token_id=end_id, pad_token_id=pad_id,
		                    compact_ids=compact_ids, compact_map=cmap, compact_unmap=unmap, fixed_token_length=fixed_token_length, token_length=token_length,
		                    use_masks=use_masks)

	def configure_target(self, target_config: TargetConfig, target_vocab: Sequence[str]):
		self.target_config = target_config
		self.target_vocab = target_vocab if isinstance(target_vocab, tuple) else tuple(target_vocab)
		self.target_configuration = {k: (v.tolist() if isinstance(v, torch.Tensor) else str(v) if isinstance(v, torch.dtype) else v) for k, v in dataclasses.asdict(target_config).items()}

	# ---- target (de)tokenisation (reference :331-406) ----
	def tokenize_target(self, text: Union[str, Sequence[str]], max_tokens: Optional[int] = None) -> tuple[torch.Tensor, Optional[torch.Tensor]]:
		tc = self.target_config
		if not tc:
			raise ValueError("Must provide target configuration before tokenizing a target noun")
		out = self.tokenize(text=text, max_tokens=max_tokens, output_dict=True)
		ids = out["input_ids"]
		lo = 1 if (self.start_token_id is not None and tc.start_token_id is None) else 0
		hi = ids.shape[1] - 1 if tc.end_token_id is None else ids.shape[1]
		ids = ids[:, lo:hi]
		mask = torch.logical_not(out["attention_mask"][:, lo:hi]) if tc.use_masks else None
		if tc.compact_ids:
			if tc.end_token_id is None and mask is not None:
				mask[ids == self.end_token_id] = True
			ids = tc.compact_map[ids.long()]
			if self.start_token_id is None and tc.start_token_id is not None:
				ids = torch.cat((ids.new_ones((ids.shape[0], 1)), ids), dim=1)
				if mask is not None:
					mask = torch.cat((mask.new_zeros((mask.shape[0], 1)), mask), dim=1)
		elif tc.end_token_id is None:
			is_end = ids == self.end_token_id
			ids = ids.masked_fill(is_end, tc.pad_token_id)
			if mask is not None:
				mask[is_end] = True
		if tc.fixed_token_length:
			n = ids.shape[1]
			if n > tc.token_length:
				raise ValueError(f"Sequence length {n} is larger than the configured target tokenization fixed length {tc.token_length}")
			if n < tc.token_length:
				full = ids.new_full((ids.shape[0], tc.token_length), tc.pad_token_id)
				full[:, :n] = ids
				ids = full
				if mask is not None:
					fm = mask.new_ones((ids.shape[0], tc.token_length))
					fm[:, :n] = mask
					mask = fm
		if self.check:
			assert ids.min() >= 0 and ids.max() < tc.vocab_size
			back = self.detokenize_target(ids.squeeze(0) if isinstance(text, str) else ids)
			if back != (text if isinstance(text, str) else list(text)):
				raise ValueError("Detokenized text is not equivalent to the original text")
		return ids, mask

	def detokenize_target(self, token_ids: torch.Tensor) -> Union[str, list[str], list[list[str]]]:
		tc = self.target_config
		if not tc:
			raise ValueError("Must provide target configuration before detokenizing a target noun")
		if tc.compact_ids:
			if self.start_token_id is None and tc.start_token_id is not None:
				token_ids = token_ids[..., 1:]
			token_ids = tc.compact_unmap[token_ids.long()]
		if token_ids.ndim == 3:
			return [self.detokenize(t) for t in token_ids]
		return self.detokenize(token_ids)

	# ---- text path (reference :423-426, :557-583) ----
	def attach_text_tower(self, tower):
		"""tower: novic_amd.clip_text.NativeTextTower (or any callable B x S device token ids -> B x F fp32 unit rows)."""
		self.text_tower = tower

	def inference_tokens(self, tokens_dict: dict) -> torch.Tensor:
		"""Tokenised text (dict with 'input_ids' B x L, CPU or device) -> B x F unit embeddings on the device (reference :557-583: pad to the context
		length with the pad token, encode, fp32, normalise)."""
		assert self._inference, "inference_tokens() must be called within inference_mode()"
		tower = getattr(self, "text_tower", None)
		if tower is None:
			raise ValueError("No text tower attached: provide local CLIP text weights (see INTEGRATION.md)")
		ids = tokens_dict["input_ids"]
		if ids.device != self.device:
			ids = ids.pin_memory().to(self.device, non_blocking=True) if ids.device.type == "cpu" and self.device.type == "cuda" else ids.to(self.device)
		ctx = tower.cfg.context_length
		if ids.shape[1] > ctx:
			raise ValueError(f"Provided token sequences are longer than the context length: {ids.shape[1]} > {ctx}")
		return tower(ids)  # causal attention: positions after the end token cannot influence it, so the padding up to the context length is not computed

	def inference_text(self, text, max_tokens: Optional[int] = None) -> torch.Tensor:
		return self.inference_tokens(self.tokenize(text, max_tokens=max_tokens, output_dict=True))

	# ---- image path (reference :432, :759-764) ----
	def attach_image_tower(self, tower):
		"""tower: novic_amd.clip_vit.NativeViT (or any callable B x 3 x R x R device tensor -> B x F fp32 unit rows)."""
		self.image_tower = tower

	def get_image_transform(self, uint8: bool = False):
		"""The tower's inference preprocess (reference: embedders.py:755-757).  uint8 = True (no reference counterpart): the same resize / crop, the pixels left as
		3 x R x R uint8 -- `inference_image` takes such batches and applies ToTensor / Normalize on the device with the same fp32 arithmetic (bit-identical embeddings,
		a quarter of the bytes over PCIe)."""
		if self.image_tower is None:
			raise ValueError("No image tower attached")
		return self.image_tower.get_image_transform(uint8=True) if uint8 else self.image_tower.get_image_transform()

	def inference_image(self, images: torch.Tensor) -> torch.Tensor:
		assert self._inference, "inference_image() must be called within inference_mode()"
		if self.image_tower is None:
			raise ValueError("No image tower attached: provide local ViT weights (see INTEGRATION.md)")
		if images.device != self.device:
			if images.device.type == "cpu" and self.device.type == "cuda":
				# (reference :761 `images.pin_memory().to(..., non_blocking=True)`: a pre-pinned staging ring instead of a page-locked allocation per call -- ImageStager)
				stager = image_stager(self.device)
				main = torch.cuda.current_stream(self.device)
				dev_images, copied, slot = stager.stage(images)
				main.wait_event(copied)
				out = self.image_tower(dev_images)
				stager.release(slot, main)
				return out
			images = images.to(self.device)
		return self.image_tower(images)

	# How many CUs the tower's persistent GEMM grids take while a decoder works on the previous batch (inference_image_batches): ViT-B/32 at batch 256 + greedy decode
	# 42.6 k labels/s one after the other, 57.0 k pipelined on all 256 CUs, 60.2 k on 208 (beam-4: 37.4 k / 44.9 k / 46.5 k; tools/e2e_overlap.py)
	pipeline_cus = None  # None: pipeline_budget(rows of the tower's GEMMs) below; a number overrides it for every batch size

	# Consecutive caller batches the pipelined tower runs as ONE forward (inference_image_batches): ViT-B/32 at a caller batch of 256 fills 59 % of the chip in its
	# out-projection / fc2 GEMMs (150 tiles on 256 CUs), four such batches fill it (600 tiles); the embeddings are handed out per caller batch as before.
	coalesce_rows = 65536  # token rows per coalesced forward at most (ViT-B/32 at batch 256: 12 800 rows -> 4 batches; ViT-L/14: 65 792 -> never)
	coalesce_max = 4

	def inference_image_batches(self, batches, persistent_cus: Optional[int] = None, coalesce: Optional[int] = None, grouped: bool = False, latency: bool = False,
	                            decode_lanes: int = 1):
		"""Generator over the embeddings of consecutive image batches, PIPELINED (no reference counterpart: infer.py:642-650 embeds and decodes one batch after the other):
		see `pipeline_image_batches`.  Enters inference_mode() by itself around each tower launch (a generator must not hold that context across its yields).
		coalesce: how many consecutive batches of one shape the tower may run as one forward (None: as many as keep the forward within `coalesce_rows` token rows, at
		most `coalesce_max`; 1: never).  One embedding tensor per CALLER batch is yielded either way -- unless grouped = True: then (embeddings, [batch sizes]) per tower launch.
		WHAT COALESCING COSTS (advisor, round 5): the pipeline takes up to 2 x group batches from the caller's iterator before the first result is yielded (8 at a group of 4),
		and the host staging ring grows to 2 x group + 1 buffers per batch shape (nine pinned + nine device buffers of 154 MB for fp32 batches of 256 x 3 x 224 x 224: ~1.4 GB
		each side); a ragged last group captures one more graph per distinct shape list.  A LIVE or slow producer -- a camera, a request queue -- wants its first answer after
		one batch, not eight: `latency=True` (= coalesce 1: one batch per tower launch, two batches of look-ahead).
		WHAT IT GUARANTEES: a coalesced launch is only chosen when neither it nor the single-batch launch runs a GEMM with a K-split tail (`NativeViT.ksplit_tail_planned`: such
		tails are summed in an order that follows the launch's tile count), so every image's embedding IS the single-batch one, bit for bit, whatever the tower and batch size;
		where a tail would run, the group shrinks until none does (down to one batch per launch).
		decode_lanes: how many tower launches' embeddings the consumer decodes AT THE SAME TIME (NOVICModel.classify_image_batches: 2) -- it only selects the workgroup
		budget of the tower beside them (pipeline_budget)."""
		if self.image_tower is None:
			raise ValueError("No image tower attached: provide local ViT weights (see INTEGRATION.md)")
		if self.device.type != "cuda":
			raise ValueError("inference_image_batches() needs the 'cuda' device")
		tokens = int(getattr(getattr(self.image_tower, "cfg", None), "tokens", 50))

		def run(images):
			with self.inference_mode():
				if isinstance(images, (list, tuple)):
					return self.image_tower.forward_many(images) if len(images) > 1 else self.image_tower(images[0])
				return self.image_tower(images)
		def cus(images):
			if persistent_cus is not None:
				return int(persistent_cus)
			if self.pipeline_cus is not None:
				return int(self.pipeline_cus)
			n = sum(im.shape[0] for im in images) if isinstance(images, (list, tuple)) else images.shape[0]
			first = images[0] if isinstance(images, (list, tuple)) else images
			# (host batches: their H2D copies run beside the tower too, and lost 4-9 % when the tower kept every CU -- 74.1 k -> 67.3 k labels/s from pinned fp32 images; the
			# reservation stays for them)
			return pipeline_budget(n * tokens, decode_lanes if first.device.type != "cpu" else 1)
		def group(images):
			if latency or not hasattr(self.image_tower, "forward_many"):
				return 1
			n = max(1, int(coalesce)) if coalesce is not None else max(1, min(int(self.coalesce_max), int(self.coalesce_rows) // max(1, images.shape[0] * tokens)))
			tail = getattr(self.image_tower, "ksplit_tail_planned", None)
			if tail is None:
				return n if coalesce is not None else 1  # (a tower that cannot say whether its launches are exact is only coalesced on request)
			B = images.shape[0]
			if n > 1 and tail(B, cus(images)):
				return 1  # the single-batch launch itself sums a tail: nothing a group could be identical to
			while n > 1 and tail(n * B, cus([images] * n)):
				n -= 1
			return n
		return pipeline_image_batches(run, batches, self.device, cus, coalesce=group, grouped=grouped)


def pipeline_budget(rows: int, decode_lanes: int = 1) -> int:
	"""Workgroups for a tower's persistent GEMM grids while a decoder works beside it, by the rows of the tower's GEMMs (images x tokens).  Measured with ViT-B/32 + greedy /
	beam-4 decode (tools/e2e_overlap.py, tools/e2e_budget_sweep.py, late round 4): batch 256 (12 800 rows) 58.7 k labels/s on 256 CUs, 61.9 k on 208, 63.4 k on 184, 63.6 k on 160;
	batch 512: 66.4 / 74.9 / 73.8 / 71.4 k; batch 1 024 (51 200 rows): 80.5 / 83.4 / 78.0 / 74.5 k; ViT-L/14 at batch 256 (65 792 rows, the tower 15 x the decode): 5 805 on 256,
	5 536 on 208, 5 203 on 184 -- the more rounds of tiles the tower's GEMMs run, the more a smaller grid costs it and the less the decoder's share matters.
	decode_lanes >= 2 (round 6): the consumer decodes two tower launches' embeddings at the same time.  What a decode call costs beside a tower is not workgroups but
	latency -- every one of its ~370 dependent launches takes 2-3 x as long while the tower streams through HBM and the L2s (a 1 024-row greedy call 3.0 ms alone, 12.2 ms beside
	the tower: LONGER than the tower's 10.9 ms, so the pipeline waited for the decoder, tools/e2e_timeline.py) -- and two calls side by side hide each other's latency (8 ms per
	1 024 rows).  The reservation then buys nothing: from 40 000 rows on the tower keeps all 256 (ViT-B/32, 1 024 images per launch, long runs: 87.4 k labels/s one call at a
	time on 208; two lanes 91.6 k on 208, 92.2 k on 240 / 256; 2 048 images per launch: 90.7 k -> 96.2 k; tools/e2e_lanes_sweep.sh); below (512 images per launch: 84.3 k on
	208, 82.8 k on 256) the rule above stands."""
	if decode_lanes >= 2 and rows >= 40000:
		return 256
	return 184 if rows < 16384 else (208 if rows < 60000 else 256)


_stagers: dict = {}


class ImageStager:
	"""Host image batches -> device, without a pinned allocation per call.  The reference's `images.pin_memory().to(device, non_blocking=True)` (embedders.py:761) is free
	when the DataLoader already delivers pinned batches (classification_dataset.py:220: pin_memory=True) and otherwise page-locks a fresh 154 MB buffer per ViT-B/32
	batch of 256 -- milliseconds of driver time in front of a 3.6 ms tower.  Here: per batch shape a ring of `depth` PRE-PINNED staging buffers and `depth` device buffers, and
	a copy stream of its own.  A pinned batch is copied straight out of the caller's tensor (which, as with any non_blocking copy, must stay untouched until the copy has
	run); a pageable one goes through the next staging buffer (one host memcpy).  Events order everything: a device buffer is overwritten only after the tower that read it
	has run (`release`), a staging buffer only after its previous H2D copy has completed."""

	def __init__(self, device: torch.device, depth: int = 3, max_shapes: int = 2):
		self.device, self.depth, self.max_shapes = torch.device(device), max(2, int(depth)), max_shapes
		from . import ops
		self.copy_stream = ops.named_stream(self.device, "h2d")
		self.rings: dict = {}
		self.bytes_copied = 0

	def _ring(self, images: torch.Tensor) -> dict:
		key = (tuple(images.shape), images.dtype)
		ring = self.rings.get(key)
		if ring is None:
			if len(self.rings) >= self.max_shapes:  # (a full batch shape and a ragged last one; a third evicts the oldest, once nothing is in flight out of it)
				torch.cuda.synchronize(self.device)
				self.rings.pop(next(iter(self.rings)))
			with torch.inference_mode(False):  # (the ring outlives the caller's inference_mode(): buffers created inside it could not be written outside it later)
				ring = self.rings[key] = dict(dev=[torch.empty(images.shape, dtype=images.dtype, device=self.device) for _ in range(self.depth)], pinned=[None] * self.depth,
				                              h2d=[None] * self.depth, free=[None] * self.depth, n=0)
		return ring

	def reserve(self, images: torch.Tensor, depth: int):
		"""At least `depth` device / staging buffers for this batch shape (coalesced towers keep 2 x group + 1 batches staged): grows the ring."""
		ring = self._ring(images)
		have = len(ring["dev"])
		if depth > have:  # appended behind the existing slots (slot indices held by in-flight batches stay valid); the fresh ones are used next
			with torch.inference_mode(False):
				ring["dev"] += [torch.empty(images.shape, dtype=images.dtype, device=self.device) for _ in range(depth - have)]
			for name in ("pinned", "h2d", "free"):
				ring[name] += [None] * (depth - have)
			ring["n"] = have  # (which slot comes next is free to choose: every reuse of a slot is ordered by its own events)

	def stage(self, images: torch.Tensor):
		"""-> (device tensor, event recorded behind its H2D copy on the copy stream, slot for `release`)."""
		assert images.device.type == "cpu"
		ring = self._ring(images)
		k = ring["n"] % len(ring["dev"])
		ring["n"] += 1
		src = images.contiguous()
		if not src.is_pinned():
			if ring["h2d"][k] is not None:
				ring["h2d"][k].synchronize()  # the copy that last read staging buffer k (depth batches ago) has completed
			if ring["pinned"][k] is None:
				with torch.inference_mode(False):
					ring["pinned"][k] = torch.empty(images.shape, dtype=images.dtype).pin_memory()
			ring["pinned"][k].copy_(src)
			src = ring["pinned"][k]
		if ring["free"][k] is not None:
			self.copy_stream.wait_event(ring["free"][k])  # the tower that read device buffer k has run
		with torch.cuda.stream(self.copy_stream):
			ring["dev"][k].copy_(src, non_blocking=True)
		ev = torch.cuda.Event()
		ev.record(self.copy_stream)
		ring["h2d"][k] = ev
		self.bytes_copied += src.numel() * src.element_size()
		return ring["dev"][k], ev, (ring, k)

	@staticmethod
	def release(slot, stream):
		"""The consumer of the staged buffer has been enqueued on `stream`: the buffer may be overwritten behind it."""
		ring, k = slot
		ev = torch.cuda.Event()
		ev.record(stream)
		ring["free"][k] = ev


def image_stager(device: torch.device) -> ImageStager:
	device = torch.device(device)
	st = _stagers.get(device)
	if st is None:
		st = _stagers[device] = ImageStager(device)
	return st


def pipeline_image_batches(tower, batches, device: torch.device, persistent_cus=None, ahead: int = 1, coalesce=1, grouped: bool = False):
	"""Generator: tower(images) for consecutive image batches, the tower of batch i + 1 enqueued on a stream of its own BEFORE batch i's embeddings are handed out, so whatever
	the consumer enqueues for batch i on its stream -- the decoder -- runs beside it.  The tower's persistent GEMM grids are launched on `persistent_cus` CUs meanwhile
	(`ops.cu_budget`: a per-call argument of the C ABI, no process-wide switch is touched): the decode step's small kernels find free CUs instead of waiting for a whole grid to end.
	HOST batches (the reference's interface: `inference_image` takes CPU images, embedders.py:759-764) travel through `ImageStager` on a copy stream, TWO tower launches ahead: the
	H2D copy of batch i + 2 runs under the tower of batch i + 1 and the decoding of batch i.
	coalesce (a number, or a function of the first image batch of a group): up to that many CONSECUTIVE batches of one shape and dtype become one tower launch -- tower is then
	called with the LIST of batches and returns the embeddings of all their rows (NativeViT.forward_many) -- and the embeddings are still yielded per caller batch, in order
	(grouped = True: per tower launch instead, as (embeddings of all its rows, [batch sizes]) -- for a consumer that decodes several caller batches at once).
	The embeddings equal tower(images) called directly
	unless one of its GEMMs runs a K-split tail (those are planned per round of that many tiles: last-bit differences).  persistent_cus: a number or a function of the
	image batch / list of batches (None: pipeline_budget over the rows at 50 tokens per image -- ViT-B/32; callers with another tower pass their own, as Embedder.inference_image_batches does).  Consume it from one thread, on one stream, and do NOT call the tower directly while the generator is active: the look-ahead launch works in the same
	per-shape workspace on the side stream (closing the generator joins the side stream, after which direct calls are safe again).
	ahead: towers kept in flight beyond the batch handed out (1: the tower of batch i + 1 beside the consumer's work on batch i).  A consumer that takes `n` batches before it
	works on them -- decoding them as n concurrent lanes, `generate_many` -- passes ahead = n, so that the towers of the NEXT group are enqueued before it starts."""
	from . import ops
	import collections
	device = torch.device(device)
	main = torch.cuda.current_stream(device)
	side = ops.named_stream(device, "tower")
	stager = image_stager(device)
	it = iter(batches)
	staged = collections.deque()   # (images, event behind which they are on the device, stager slot | None, group size wanted by the batch)

	ahead = max(1, int(ahead))
	flying = collections.deque()  # towers enqueued, embeddings not handed out yet: (embeddings of the whole group, event, batch sizes)
	exhausted = [False]

	def take():
		if exhausted[0]:
			return None
		try:
			return next(it)
		except StopIteration:
			exhausted[0] = True
			return None

	def want(images) -> int:
		return max(1, int(coalesce(images) if callable(coalesce) else coalesce))

	def fill():  # keep two tower launches' worth of batches staged ahead of the tower that is launched next
		target = 2 * (staged[0][3] if staged else 1)
		while len(staged) < target:
			images = take()
			if images is None:
				return
			n = want(images)
			if images.device.type == "cpu":
				stager.reserve(images, 2 * n + 1)
				staged.append(stager.stage(images) + (n,))
			else:
				# a device batch: whatever produced it is on the consumer's stream NOW -- an event here, two batches ahead of the tower's launch, instead of a wait for the
				# stream's tail at launch time (which by then carries the decoding of earlier batches: a false dependency that serialises towers behind decode lanes)
				images = images if images.device == device else images.to(device)
				ready = torch.cuda.Event()
				ready.record(main)
				staged.append((images, ready, None, n))
			target = 2 * staged[0][3]

	def launch():
		first = staged[0]
		group = [staged.popleft()]
		while staged and len(group) < first[3] and staged[0][0].shape == first[0].shape and staged[0][0].dtype == first[0].dtype:
			group.append(staged.popleft())
		for _, copied, _, _ in group:
			side.wait_event(copied)  # the batch is on the device: its H2D copy (copy stream) / whatever the consumer's stream held when it was taken from the iterator
		arg = [g[0] for g in group] if len(group) > 1 else group[0][0]
		with ops.cu_budget(int(persistent_cus(arg)) if callable(persistent_cus) else (pipeline_budget(sum(g[0].shape[0] for g in group) * 50) if persistent_cus is None else int(persistent_cus))), torch.cuda.stream(side):
			e = tower(arg)
			for images, _, slot, _ in group:
				if slot is not None:
					stager.release(slot, side)
				else:
					images.record_stream(side)
		ev = torch.cuda.Event()
		ev.record(side)
		return e, ev, [g[0].shape[0] for g in group]
	def top_up():  # `ahead` towers in flight beyond the one about to be handed out
		fill()
		while staged and len(flying) < ahead + 1:
			flying.append(launch())
			fill()
	try:
		top_up()
		while flying:
			e, ev, sizes = flying.popleft()
			top_up()
			main.wait_event(ev)
			e.record_stream(main)
			if grouped:
				yield e, sizes
			elif len(sizes) == 1:
				yield e
			else:
				row = 0
				for n in sizes:
					yield e[row:row + n]
					row += n
	finally:
		# The consumer stopped early (break, exception, generator close) with the look-ahead tower -- and a copy behind it -- still in flight: join them, so that whatever
		# runs next on the consumer's stream (a direct tower call reuses the same per-shape workspace and graph output) is ordered behind them.
		main.wait_stream(side)
		main.wait_stream(stager.copy_stream)


class LocalVocabEmbedder(Embedder):
	"""Self-contained embedder over an explicit token list: text = space-separated tokens, ids = indices into the list.

	Special tokens follow the CLIP tokenizer conventions the reference relies on: one start token, one end token, pad = a
	dedicated id (0 here) that never appears as content.
	"""

	def __init__(self, tokens: Sequence[str], embed_dim: int, context_length: int = 77, token_dtype: torch.dtype = torch.int64, with_start: bool = True, **kwargs):
		self.tokens = list(tokens)
		specials = ["<pad>", "<start>", "<end>"] if with_start else ["<pad>", "<end>"]
		self.itos = specials + self.tokens
		self.stoi = {t: i for i, t in enumerate(self.itos)}
		if len(self.stoi) != len(self.itos):
			raise ValueError("Token list contains duplicates or special-token names")
		kwargs.setdefault("load_model", False)
		super().__init__(configuration=dict(type="local", num_tokens=len(self.tokens), embed_dim=embed_dim, context_length=context_length, with_start=with_start),
		                 context_length=context_length, vocab_size=len(self.itos), cased_tokens=True, start_token_id=(1 if with_start else None),
		                 end_token_id=(2 if with_start else 1), pad_token_id=0, token_dtype=token_dtype, embed_dtype=torch.float32, embed_dim=embed_dim, **kwargs)

	@classmethod
	def from_file(cls, path: str, **kwargs) -> "LocalVocabEmbedder":
		with open(path, "r") as f:
			spec = json.load(f)
		kwargs.pop("amp", None)
		kwargs.pop("amp_bf16", None)
		return cls(tokens=spec["tokens"], embed_dim=spec["embed_dim"], context_length=spec.get("context_length", 77), with_start=spec.get("with_start", True), **kwargs)

	def tokenize(self, text, max_tokens=None, output_dict=False):
		texts = (text,) if isinstance(text, str) else tuple(text)
		rows = []
		for t in texts:
			ids = [self.stoi[w] for w in t.split()] if t else []
			row = ([self.start_token_id] if self.start_token_id is not None else []) + ids + [self.end_token_id]
			rows.append(row)
		L = max(len(r) for r in rows)
		if max_tokens is not None:
			L = max(L, 0) if L <= max_tokens else max_tokens
		if L > self.context_length:
			raise ValueError(f"Tokenization is longer than the context length {self.context_length}")
		ids = torch.full((len(rows), L), self.pad_token_id, dtype=self.token_dtype)
		att = torch.zeros((len(rows), L), dtype=torch.int64)
		for i, r in enumerate(rows):
			r = r[:L]
			ids[i, :len(r)] = torch.tensor(r, dtype=self.token_dtype)
			att[i, :len(r)] = 1
		return {"input_ids": ids, "attention_mask": att} if output_dict else ids

	def detokenize(self, token_ids: torch.Tensor):
		def one(row):
			words = []
			for t in row.tolist():
				if t == self.start_token_id:
					continue
				if t in (self.end_token_id, self.pad_token_id):
					break
				words.append(self.itos[t])
			return " ".join(words)
		return one(token_ids) if token_ids.ndim == 1 else [one(r) for r in token_ids]


class TransformersEmbedder(Embedder):
	"""The reference's TransformersEmbedder (embedders.py:767-907) for a LOCAL Hugging Face CLIP directory (config.json, the tokenizer files, and
	model.safetensors or pytorch_model.bin): spec 'transformers:/path/to/dir'.

	Host side as in the reference: transformers' own tokenizer classes (loaded with local_files_only), the same special-token bookkeeping, the same
	tokenize / detokenize calls.  Device side replaced: instead of AutoModel the directory's weights are loaded into the native towers
	(clip_vit.NativeViT, clip_text.NativeTextTower -- hand-written HIP kernels), which return the unit-norm fp32 embeddings of
	get_text_features / get_image_features + F.normalize (:890, :906-907).  Hub names are refused: this build never touches the network.
	"""

	def __init__(self, model_id: str, amp: bool = True, amp_bf16: bool = False, tokenizer_batch_size: int = 1024, inference_batch_size: int = 256, image_batch_size: int = 128,
	             load_model: bool = True, compile_model: bool = False, use_optimum: bool = False, device: Union[int, str, torch.device] = "cuda", check: bool = False):
		if not os.path.isdir(model_id):
			raise ValueError(f"TransformersEmbedder needs a local model directory (no network access in this build): {model_id}")
		import transformers
		self.model_id = model_id
		self.use_optimum = use_optimum  # accepted for signature parity; the native towers have no BetterTransformer variant
		self.config = transformers.AutoConfig.from_pretrained(model_id, local_files_only=True)
		self.model_type = self.config.model_type.upper()
		if self.model_type != "CLIP":
			raise ValueError(f"The native towers implement the CLIP architecture only, not {self.model_type}")
		for sub in ("text_config", "vision_config"):  # reference :790-799: some released configs carry a wrong projection_dim in the sub-configs
			subcfg = getattr(self.config, sub, None)
			if subcfg is not None and getattr(subcfg, "projection_dim", self.config.projection_dim) != self.config.projection_dim:
				subcfg.projection_dim = self.config.projection_dim
		self.tokenizer = transformers.AutoTokenizer.from_pretrained(model_id, local_files_only=True)
		tk = self.tokenizer
		start_token_id = tk.bos_token_id if tk.bos_token_id is not None else tk.cls_token_id
		end_token_id = tk.eos_token_id if tk.eos_token_id is not None else tk.sep_token_id
		end_token = tk.eos_token if tk.eos_token_id is not None else tk.sep_token
		pad_token_id, pad_token = tk.pad_token_id, tk.pad_token
		pad_aliases = {token for token, token_id in tk.get_vocab().items() if token_id == pad_token_id}
		pad_aliases.discard(pad_token)
		pad_aliases.discard(end_token)
		if pad_aliases:
			raise ValueError(f"Pad token {pad_token_id} cannot have non-end token aliases: {pad_aliases}")
		self.text_tower = None
		self.image_processor = None
		super().__init__(configuration={"model_id": self.model_id, "model_config": format(self.config)}, context_length=tk.model_max_length, vocab_size=len(tk),
		                 cased_tokens=(tk.encode("CPU") != tk.encode("cpu")), start_token_id=start_token_id, end_token_id=end_token_id, pad_token_id=pad_token_id,
		                 token_dtype=torch.int64, embed_dtype=torch.float32, embed_dim=self.config.projection_dim,
		                 amp_mode=False if not amp else torch.bfloat16 if amp_bf16 else True, manual_amp_dtype=None, tokenizer_batch_size=tokenizer_batch_size,
		                 inference_batch_size=inference_batch_size, image_batch_size=image_batch_size, load_model=load_model, compile_model=compile_model, device=device, check=check)

	def _read_weights(self) -> dict:
		st = os.path.join(self.model_id, "model.safetensors")
		if os.path.isfile(st):
			from safetensors.torch import load_file
			return load_file(st)
		pt = os.path.join(self.model_id, "pytorch_model.bin")
		if os.path.isfile(pt):
			return torch.load(pt, map_location="cpu", weights_only=True)
		raise ValueError(f"No model.safetensors or pytorch_model.bin in {self.model_id}")

	def load_model(self) -> bool:
		if self.is_model_loaded():
			return False
		from . import clip_text, clip_vit
		hf = {k: v.float() for k, v in self._read_weights().items()}
		vc, tc, F = self.config.vision_config, self.config.text_config, self.config.projection_dim
		vit = clip_vit.NativeViT(clip_vit.ViTConfig(image_size=vc.image_size, patch_size=vc.patch_size, width=vc.hidden_size, layers=vc.num_hidden_layers, heads=vc.num_attention_heads,
		                                            mlp_ratio=vc.intermediate_size / vc.hidden_size, embed_dim=F, quick_gelu=(vc.hidden_act == "quick_gelu"), ln_eps=vc.layer_norm_eps))
		vit.load_hf_state_dict(hf)
		# transformers pools at the first END token (eos_token_id), except for the legacy eos_token_id == 2 configs, which pool at the arg-max id
		eot = None if tc.eos_token_id == 2 else int(tc.eos_token_id)
		txt = clip_text.NativeTextTower(clip_text.TextConfig(vocab_size=tc.vocab_size, context_length=tc.max_position_embeddings, width=tc.hidden_size, layers=tc.num_hidden_layers,
		                                                     heads=tc.num_attention_heads, mlp_ratio=tc.intermediate_size / tc.hidden_size, embed_dim=F,
		                                                     quick_gelu=(tc.hidden_act == "quick_gelu"), ln_eps=tc.layer_norm_eps), eot_token_id=eot)
		txt.load_hf_state_dict(hf)
		self.image_tower, self.text_tower = vit.to(self.device), txt.to(self.device)
		try:
			import transformers
			self.image_processor = transformers.AutoImageProcessor.from_pretrained(self.model_id, local_files_only=True)
		except Exception:  # no preprocessor_config.json (or no vision extras installed): the tower's own CLIP preprocessing, which is the same recipe
			self.image_processor = None
		return True

	def unload_model(self) -> bool:
		if not self.is_model_loaded():
			return False
		self.image_tower = self.text_tower = self.image_processor = None
		return True

	def is_model_loaded(self) -> bool:
		return self.image_tower is not None and self.text_tower is not None

	def tokenize(self, text, max_tokens: Optional[int] = None, output_dict: bool = False):
		out = self.tokenizer(text=text, padding=True, truncation=True, max_length=max_tokens, return_tensors="pt")  # reference :870
		return dict(out) if output_dict else out["input_ids"]

	def detokenize(self, token_ids: torch.Tensor):
		if token_ids.ndim <= 1:
			return self.tokenizer.decode(token_ids, skip_special_tokens=True)
		return self.tokenizer.batch_decode(token_ids, skip_special_tokens=True)

	def get_image_transform(self):
		if self.image_processor is None:
			return super().get_image_transform()
		return lambda image: self.image_processor(images=image, return_tensors="pt")["pixel_values"].squeeze(dim=0)  # reference :897-900


def __getattr__(name):  # embedders.OpenCLIPEmbedder / embedders.OpenAIEmbedder, as in the reference's module (defined in local_clip.py, which imports this module)
	if name in ("OpenCLIPEmbedder", "OpenAIEmbedder"):
		from . import local_clip
		return getattr(local_clip, name)
	raise AttributeError(f"module {__name__!r} has no attribute {name!r}")
