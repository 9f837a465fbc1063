"""Embedding-cache reader and writer: the binary file that feeds the training loop.

Reads the reference's cache format bit-for-bit (reference embedding_cache.py:24-31 layout, ``Header`` :34-73 with struct
``'<32sB?????32s32sLLHHHLHHHH'`` = 128 bytes, ``Meta`` :76-154 offsets) and reproduces ``EmbeddingCache`` (:471-756: validation,
``get_samples``) and ``EmbeddingCache.Dataset.__getitem__`` (:827-895: contiguous batch slices with per-epoch rotation and wrap, multi-target
trimming, weight renormalisation, trailing-padding column trimming, ``multi_first`` transposition).

MI355X-first loader (``DeviceLoader``): instead of forked DataLoader workers + pinned copies per batch (reference :918-958), the token table,
the per-embedding target ids / weights and the embedding vectors are uploaded to HBM ONCE (a 2 M x 512 fp32 cache is 4 GB of 288 GB) and
every batch is assembled on the device by one gather kernel (``novic_cache_gather``); the data-dependent trims (longest label in the batch,
number of non-empty targets) are decided on the host from per-noun lengths precomputed at load time, so no device synchronisation happens
per batch.  A cache beyond the HBM budget streams its embedding rows -- the next few batches' contiguous file slices -- through pinned staging buffers
and a copy stream into device slabs (``DeviceLoader(stream_depth=...)``), the small tables staying resident; batches are identical either way.

Writer (``EmbeddingCacheWriter``, reference :161-459; ``RandomCacheWriter`` / ``PhotoCacheWriter``, embedding_cache_writers.py:23-104): the same file,
byte for byte, from the same inputs -- pinned by writing one cache with the reference's writer and one with this one (tests/golden/make_golden_cache.py).
"""
from __future__ import annotations

import contextlib
import dataclasses
import itertools
import mmap
import os
import random
import struct
from typing import Iterator, Optional, Sequence

import numpy as np
import torch

from . import embedders, embedding_dataset, ops


@dataclasses.dataclass(frozen=True)
class Header:
	VERSION = 1
	MAGIC_SIZE = 32
	MAGIC_BYTES = b"\xa9\xfdK\x14*\x9a\xb8\x13m\x157\xca\xe8+\xef\x82B\x19\xdbJ\xb8\x93\xb2&\xa0\x1a=\xe4\xadR\xb1\x99"
	STRUCT = struct.Struct("<32sB?????32s32sLLHHHLHHHH")
	INT_DTYPES = (torch.int8, torch.int16, torch.int32, torch.int64)
	BOOL_DTYPES = (torch.bool,)
	FLOAT_DTYPES = (torch.float16, torch.bfloat16, torch.float32, torch.float64)

	magic_bytes: bytes
	version: int
	use_targets: bool
	full_targets: bool
	default_weights: bool
	unit_weights: bool
	embedder_strict: bool
	embedder_hash: bytes
	target_config_hash: bytes
	target_nouns_num: int
	target_nouns_size: int
	target_dim: int
	target_dtype_id: int
	target_mask_dtype_id: int
	embed_num: int
	embed_targets_dim: int
	embed_targets_dtype_id: int
	embed_dim: int
	embed_dtype_id: int


assert Header.STRUCT.size == 128

_NP = {torch.int8: np.int8, torch.int16: np.int16, torch.int32: np.int32, torch.int64: np.int64, torch.bool: np.bool_, torch.float16: np.float16, torch.float32: np.float32,
       torch.float64: np.float64}


@dataclasses.dataclass(frozen=True)
class Meta:
	target_dtype: torch.dtype
	target_mask_dtype: torch.dtype
	embed_targets_dtype: torch.dtype
	embed_dtype: torch.dtype
	target_nouns_offset: int
	target_offset: int
	target_mask_offset: int
	embed_targets_offset: int
	embed_target_weights_offset: int
	embed_offset: int
	total_size: int

	@classmethod
	def from_header(cls, h: Header) -> "Meta":
		td, md, ed, fd = Header.INT_DTYPES[h.target_dtype_id], Header.BOOL_DTYPES[h.target_mask_dtype_id], Header.INT_DTYPES[h.embed_targets_dtype_id], Header.FLOAT_DTYPES[h.embed_dtype_id]
		size = lambda dt: torch.tensor((), dtype=dt).element_size()
		off_nouns = Header.STRUCT.size
		off_target = off_nouns + h.target_nouns_size
		off_mask = off_target + h.target_nouns_num * h.target_dim * size(td)
		off_ids = off_mask + h.target_nouns_num * h.target_dim * size(md)
		off_w = off_ids + h.embed_num * h.embed_targets_dim * size(ed)
		off_embed = off_w + h.embed_num * h.embed_targets_dim * size(fd)
		return Meta(td, md, ed, fd, off_nouns, off_target, off_mask, off_ids, off_w, off_embed, off_embed + h.embed_num * h.embed_dim * size(fd))


class EmbeddingCacheWriter:
	"""Writes the cache file format the reader below (and the reference's reader) consume: reference embedding_cache.py:161-459, same constructor
	arguments, same header bytes, same validation of what is written, the magic bytes only once everything has arrived.  Host-side file I/O: the
	embeddings it is fed come from the native towers (PhotoCacheWriter) or from the caller."""

	INIT_MAGIC_BYTES = b"\x00" * Header.MAGIC_SIZE
	TARGET_EXCLUDE = {"fixed_token_length"}  # reference :45: fields of TargetConfig that do not affect whether a cache can be used

	def __init__(self, cache_path: str, embedder: embedders.Embedder, num_embed: int, shuffle: bool = True, use_targets: bool = True, full_targets: bool = True,
	             target_nouns: Optional[Sequence[str]] = None, num_embed_targets: int = 1, default_weights: bool = False, unit_weights: bool = True, embedder_strict: bool = True):
		self.use_targets = use_targets
		self.cache_path = os.path.abspath(cache_path)
		self.embedder = embedder
		self.num_embed = num_embed
		self.shuffle = shuffle
		self.num_embed_targets = num_embed_targets if use_targets else 0
		self.full_targets = full_targets or not use_targets or self.num_embed_targets <= 1
		self.default_weights = default_weights or not use_targets
		self.unit_weights = unit_weights or self.default_weights
		self.embedder_strict = embedder_strict
		if not use_targets:
			self.target_nouns = ()
		elif target_nouns is None:
			raise ValueError("Target nouns must be provided if use_targets=True")
		else:
			self.target_nouns = ("",) + tuple(target_nouns)  # ID 0 = the fully padded target
		self.num_target_nouns = len(self.target_nouns)
		self.target_noun_map = {noun: i for i, noun in enumerate(self.target_nouns)}
		if len(self.target_noun_map) != (self.num_target_nouns - 1 if "" in self.target_nouns[1:] else self.num_target_nouns):
			raise ValueError("There are duplicate non-empty target nouns")
		self.target_nouns_bytes = "\x00".join(self.target_nouns).encode("utf-8")
		self.embed_targets_dtype = torch.int32
		tc = embedder.target_config
		digest = lambda **kw: embedder.get_configuration_hash(hexdigest=False, algorithm="sha256", **kw)
		self.header = Header(
			magic_bytes=self.INIT_MAGIC_BYTES, version=Header.VERSION, use_targets=self.use_targets, full_targets=self.full_targets, default_weights=self.default_weights,
			unit_weights=self.unit_weights, embedder_strict=self.embedder_strict,
			embedder_hash=digest(main_config=True, target_config=False) if self.embedder_strict else b"\x00" * 32,
			target_config_hash=digest(main_config=False, target_config=True, target_exclude=self.TARGET_EXCLUDE) if self.use_targets and self.embedder_strict else b"\x00" * 32,
			target_nouns_num=self.num_target_nouns, target_nouns_size=len(self.target_nouns_bytes), target_dim=tc.token_length if self.use_targets else 0,
			target_dtype_id=Header.INT_DTYPES.index(embedder.token_dtype), target_mask_dtype_id=Header.BOOL_DTYPES.index(tc.mask_dtype if self.use_targets else torch.bool),
			embed_num=self.num_embed, embed_targets_dim=self.num_embed_targets, embed_targets_dtype_id=Header.INT_DTYPES.index(self.embed_targets_dtype),
			embed_dim=embedder.embed_dim, embed_dtype_id=Header.FLOAT_DTYPES.index(embedder.embed_dtype))
		if self.header.embed_num < 1:
			raise ValueError(f"Cache file must have a positive number of embeddings: {self.header.embed_num}")
		if self.use_targets and (self.header.target_dim < 1 or self.header.embed_targets_dim < 1):
			raise ValueError(f"Cache file must have positive target dimensions: {self.header.target_dim} token ids, {self.header.embed_targets_dim} targets per embedding")
		self.header_bytes = Header.STRUCT.pack(*dataclasses.astuple(self.header))
		self.meta = Meta.from_header(self.header)
		self.embed_eps = float(torch.finfo(self.meta.embed_dtype).eps)
		self.embed_written = self.bytes_written = 0
		self.shuffle_perm = self.target_token_ids = self.target_mask = self.cache_fd = self.default_weights_tensor = None

	def tensorize_embed_targets(self, embed_targets_str) -> torch.Tensor:
		"""Target nouns (or sequences of them) -> zero-padded B x M tensor of target noun IDs (reference :251-266)."""
		if not self.use_targets:
			raise ValueError("Cannot tensorize embedding target noun IDs if not using targets")
		out = torch.zeros((len(embed_targets_str), self.header.embed_targets_dim), dtype=self.meta.embed_targets_dtype)
		for i, targets in enumerate(embed_targets_str):
			if isinstance(targets, str):
				out[i, 0] = self.target_noun_map[targets]
			else:
				for j, target in enumerate(targets):
					out[i, j] = self.target_noun_map[target]
		return out

	def _strides(self):
		esz = lambda dt: torch.tensor((), dtype=dt).element_size()
		h, m = self.header, self.meta
		return h.embed_targets_dim * esz(m.embed_targets_dtype), h.embed_targets_dim * esz(m.embed_dtype), h.embed_dim * esz(m.embed_dtype)

	def __enter__(self) -> "EmbeddingCacheWriter":
		h, m = self.header, self.meta
		self.embed_written = self.bytes_written = 0
		self.shuffle_perm = torch.randperm(h.embed_num, dtype=torch.int32) if self.shuffle else None
		self.target_token_ids = self.target_mask = self.cache_fd = self.default_weights_tensor = None
		try:
			if self.use_targets:
				ids, mask = self.embedder.tokenize_target(text=self.target_nouns)
				if mask is None:
					mask = torch.zeros_like(ids, dtype=m.target_mask_dtype)
				ids[0, :].fill_(self.embedder.target_config.pad_token_id)
				mask[0, :].fill_(True)
				ids, mask = ids.contiguous(), mask.contiguous()
				if ids.dtype != m.target_dtype or tuple(ids.shape) != (h.target_nouns_num, h.target_dim):
					raise ValueError(f"Unexpected target token IDs tensor: Shape {tuple(ids.shape)}, DType {ids.dtype}")
				if mask.dtype != m.target_mask_dtype or tuple(mask.shape) != (h.target_nouns_num, h.target_dim):
					raise ValueError(f"Unexpected target token padding mask tensor: Shape {tuple(mask.shape)}, DType {mask.dtype}")
				self.target_token_ids, self.target_mask = ids, mask
			self.cache_fd = os.open(self.cache_path, os.O_RDWR | os.O_CREAT, 0o644)
			os.ftruncate(self.cache_fd, 0)
			os.ftruncate(self.cache_fd, m.total_size)
			self._write(self.header_bytes, 0, Header.STRUCT.size)
			if self.use_targets:
				self._write(self.target_nouns_bytes, m.target_nouns_offset, h.target_nouns_size)
				self._write(memoryview(self.target_token_ids.numpy()), m.target_offset, m.target_mask_offset - m.target_offset)
				self._write(memoryview(self.target_mask.numpy()), m.target_mask_offset, m.embed_targets_offset - m.target_mask_offset)
				if self.default_weights:
					if self.full_targets:
						w = torch.full((h.embed_num, h.embed_targets_dim), 1 / h.embed_targets_dim, dtype=m.embed_dtype)
						self._write(memoryview(w.numpy()), m.embed_target_weights_offset, m.embed_offset - m.embed_target_weights_offset)
					else:
						self.default_weights_tensor = (torch.tril(torch.ones(h.embed_targets_dim, h.embed_targets_dim)) / torch.arange(1, h.embed_targets_dim + 1).unsqueeze(1)).to(m.embed_dtype)
		except BaseException:
			fd, self.cache_fd = self.cache_fd, None
			self.target_token_ids = self.target_mask = self.default_weights_tensor = None
			if fd is not None:
				os.close(fd)
				self.remove()
			raise
		return self

	def _put(self, t: torch.Tensor, base: int, stride: int, first: int, indices):
		buf = memoryview(t.contiguous().numpy())
		if indices is None:
			self._write(buf, base + first * stride, t.shape[0] * stride)
		else:
			for i, index in enumerate(indices):
				self._write(buf[i:i + 1], base + index * stride, stride)

	def write(self, embeds: torch.Tensor, embed_targets: Optional[torch.Tensor] = None, embed_target_weights: Optional[torch.Tensor] = None):
		"""B x F unit-norm embeddings (CPU) [+ B x M target noun IDs, non-zero IDs first] [+ B x M weights, descending]: reference :329-418."""
		h, m = self.header, self.meta
		B = embeds.shape[0]
		if (embed_targets is not None) != self.use_targets:
			raise ValueError("Embedding target noun IDs were provided although none were expected, or vice versa")
		if (embed_target_weights is None) != self.default_weights:
			raise ValueError("Embedding target noun weights were provided although none were expected, or vice versa")
		if embeds.ndim != 2 or B < 1 or embeds.shape[1] != h.embed_dim or embeds.dtype != m.embed_dtype:
			raise ValueError(f"Unexpected embeddings tensor: Shape {tuple(embeds.shape)}, DType {embeds.dtype}")
		first = self.embed_written
		self.embed_written += B
		if self.embed_written > h.embed_num:
			raise ValueError(f"Invalid embedding index {first} to write {B} samples to due to the total number of embeddings only being {h.embed_num}")
		if torch.any((torch.linalg.vector_norm(embeds, dim=1) - 1).abs() > 4 * self.embed_eps):
			raise ValueError("Embeddings must always be unit vectors")
		ids_stride, w_stride, e_stride = self._strides()
		indices = self.shuffle_perm[first:self.embed_written].tolist() if self.shuffle else None
		self._put(embeds, m.embed_offset, e_stride, first, indices)
		if embed_targets is not None:
			if tuple(embed_targets.shape) != (B, h.embed_targets_dim) or embed_targets.dtype != m.embed_targets_dtype:
				raise ValueError(f"Unexpected embedding target noun IDs tensor: Shape {tuple(embed_targets.shape)}, DType {embed_targets.dtype}")
			lo, hi = torch.aminmax(embed_targets)
			if lo < 0 or hi >= self.num_target_nouns:
				raise ValueError(f"Target noun IDs tensor has values outside the expected range: IDs {lo.item()} to {hi.item()} seen given {self.num_target_nouns} target nouns")
			if self.full_targets:
				if lo <= 0:
					raise ValueError("Embedding target cannot have any zeros if full targets is specified")
			elif embed_targets[:, 0].min() <= 0:
				raise ValueError("First target must always be non-zero even if not using full targets")
			nonzero = embed_targets.bool()
			if embed_targets.shape[1] > 1 and not torch.equal(nonzero.cummin(dim=1)[0], nonzero):
				raise ValueError("All non-zero target noun IDs must come before any trailing zeros")
			self._put(embed_targets, m.embed_targets_offset, ids_stride, first, indices)
			if embed_target_weights is None and not self.full_targets:
				embed_target_weights = self.default_weights_tensor[nonzero[:, 1:].sum(dim=1)]
		if embed_target_weights is not None:
			w = embed_target_weights
			if tuple(w.shape) != (B, h.embed_targets_dim) or w.dtype != m.embed_dtype:
				raise ValueError(f"Unexpected embedding target noun weights tensor: Shape {tuple(w.shape)}, DType {w.dtype}")
			if torch.any(w < 0):
				raise ValueError("Embedding target noun weights must be non-negative")
			if w[:, 0].min() <= 0:
				raise ValueError("First target weight must always be non-zero")
			if h.embed_targets_dim > 1 and torch.any(w[:, 1:] - w[:, :-1] > 4 * self.embed_eps):
				raise ValueError("Embedding target noun weights must be in descending order")
			wnz = w.bool()
			if ((embed_targets == 0) & wnz).any():
				raise ValueError("Zero target noun IDs must have zero weight")
			if w.shape[1] > 1 and not torch.equal(wnz.cummin(dim=1)[0], wnz):
				raise ValueError("All non-zero target noun weights must come before any trailing zeros")
			if self.unit_weights and torch.any((w.sum(dim=1) - 1).abs() > 4 * self.embed_eps):
				raise ValueError("As unit weights was specified, the target noun weights are expected to sum to 1 for each embedding")
			self._put(w, m.embed_target_weights_offset, w_stride, first, indices)

	def _write(self, buffer, offset: int, expected_size: int):
		n = buffer.nbytes if isinstance(buffer, memoryview) else len(buffer)
		written = os.pwrite(self.cache_fd, buffer, offset)
		self.bytes_written += written
		if written != n:
			raise OSError(f"Failed to write all bytes in the buffer: {written} vs {n}")
		if written != expected_size:
			raise OSError(f"Written buffer was not of the expected size: {written} vs {expected_size}")

	def __exit__(self, exc_type, exc_val, exc_tb) -> bool:
		valid = False
		try:
			if exc_type is None and self.embed_written == self.header.embed_num and self.bytes_written == self.meta.total_size:
				self._write(Header.MAGIC_BYTES, 0, Header.MAGIC_SIZE)  # only a complete file gets its magic bytes
				os.fsync(self.cache_fd)
				valid = os.pread(self.cache_fd, Header.MAGIC_SIZE, 0) == Header.MAGIC_BYTES and os.fstat(self.cache_fd).st_size == self.meta.total_size
		finally:
			fd, self.cache_fd = self.cache_fd, None
			self.embed_written = self.bytes_written = 0
			self.shuffle_perm = self.target_token_ids = self.target_mask = self.default_weights_tensor = None
			os.close(fd)
			if not valid:
				self.remove()
				if exc_type is None:
					raise RuntimeError("Failed to write embedding cache")
		return False

	def remove(self):
		with contextlib.suppress(FileNotFoundError):
			os.remove(self.cache_path)


class RandomCacheWriter(EmbeddingCacheWriter):
	"""Random unit vectors without targets (reference embedding_cache_writers.py:23-47): the input recipe of the headline benchmark."""

	def __init__(self, cache_path: str, embedder: embedders.Embedder, num_embed: int, batch_size: int = 2048):
		self.batch_size = batch_size
		super().__init__(cache_path=cache_path, embedder=embedder, num_embed=num_embed, shuffle=False, use_targets=False, embedder_strict=False)

	def generate(self):
		with self:
			left = self.header.embed_num
			while left > 0:
				embeds = torch.nn.functional.normalize(torch.randn(min(self.batch_size, left), self.header.embed_dim, dtype=self.meta.embed_dtype), dim=-1)
				self.write(embeds=embeds)
				left -= embeds.shape[0]


class PhotoCacheWriter(EmbeddingCacheWriter):
	"""One embedding per target noun from the prompt 'a photo of a NOUN' (reference embedding_cache_writers.py:50-104), the text going through the
	embedder's tokenizer and the native text tower (Embedder.inference_text)."""

	def __init__(self, cache_path: str, embedder: embedders.Embedder, target_nouns: Sequence[str], debug: bool = False, shuffle: bool = True):
		self.debug = debug
		super().__init__(cache_path=cache_path, embedder=embedder, num_embed=len(target_nouns), shuffle=shuffle, use_targets=True, full_targets=True, target_nouns=target_nouns,
		                 num_embed_targets=1, default_weights=True, unit_weights=True)

	def generate(self):
		with self.embedder.inference_model(), self:
			all_embeds = torch.full((self.num_embed, self.embedder.embed_dim), float("nan"), dtype=self.embedder.embed_dtype) if self.debug else None
			all_targets = torch.arange(1, self.num_target_nouns, dtype=self.meta.embed_targets_dtype).unsqueeze(1)
			count = 0
			it = itertools.islice(self.target_nouns, 1, None)  # skip the empty string of ID 0
			while nouns := tuple(itertools.islice(it, self.embedder.inference_batch_size)):
				with self.embedder.inference_mode():
					embeds = self.embedder.inference_text(text=tuple(f"a photo of a {noun}" for noun in nouns))
				embeds = embeds.cpu()
				if self.debug:
					assert torch.equal(all_targets[count:count + len(nouns)], self.tensorize_embed_targets(nouns))
					all_embeds[count:count + len(nouns)] = embeds
				self.write(embeds=embeds, embed_targets=all_targets[count:count + len(nouns)])
				count += len(nouns)
			ret = (all_embeds, self.target_token_ids[1:].clone(), self.target_mask[1:].clone() if self.embedder.target_config.use_masks else None) if self.debug else None
		return ret


class EmbeddingCache:
	"""Validated, memory-mapped view of one cache file (context manager, like the reference's)."""

	def __init__(self, cache_path: str, embedder: embedders.Embedder, use_targets: Optional[bool] = None, strict_embedder: bool = True):
		self.cache_path = os.path.abspath(cache_path)
		self.embedder = embedder
		self.use_targets = use_targets
		self.strict_embedder = strict_embedder
		with open(self.cache_path, "rb") as f:
			self.cache_stat = os.fstat(f.fileno())
			raw = f.read(Header.STRUCT.size)
			if len(raw) != Header.STRUCT.size:
				raise ValueError(f"Cache file too short for header: {len(raw)} bytes read but {Header.STRUCT.size} needed")
			self.header_bytes = raw
			self.header = h = Header(*Header.STRUCT.unpack(raw))
			if h.magic_bytes != Header.MAGIC_BYTES:
				raise ValueError("Cache file has invalid magic bytes")
			if not 1 <= h.version <= Header.VERSION:
				raise ValueError(f"Cache file version is unsupported: {h.version} vs supported {Header.VERSION}")
			if strict_embedder and h.embedder_strict:
				if embedder.get_configuration_hash(main_config=True, target_config=False, hexdigest=False) != h.embedder_hash:
					raise ValueError("Cache file embedder hash does not match embedder hash => Incompatible (open with strict_embedder=False to check only dimension and dtype)")
			if self.use_targets is None:
				self.use_targets = h.use_targets
			if self.use_targets:
				if not h.use_targets:
					raise ValueError("Embedding cache class requires targets but the loaded cache file has none")
				if h.target_nouns_num < 1:
					raise ValueError("Cache file needs to have at least one target noun")
				self.target_nouns_bytes = f.read(h.target_nouns_size)
				if len(self.target_nouns_bytes) != h.target_nouns_size:
					raise ValueError("Cache file too short for target nouns")
				self.target_nouns = tuple(self.target_nouns_bytes.decode("utf-8").split("\x00"))
				if len(self.target_nouns) != h.target_nouns_num:
					raise ValueError(f"Cache file has an inconsistent number of target nouns: {len(self.target_nouns)} vs {h.target_nouns_num}")
				if self.target_nouns[0] != "":
					raise ValueError("First target noun in cache file must always be the empty string (which signifies 'unknown/no classification')")
			else:
				self.target_nouns_bytes, self.target_nouns = None, None
			f.seek(0, os.SEEK_END)
			self.cache_size = f.tell()
		self.meta = m = Meta.from_header(h)
		if h.embed_num < 1:
			raise ValueError(f"Cache file must have a positive number of embeddings: {h.embed_num}")
		if h.embed_dim != embedder.embed_dim:
			raise ValueError(f"Cache file has embedding dimension mismatch: {h.embed_dim} vs {embedder.embed_dim}")
		if m.embed_dtype != embedder.embed_dtype:
			raise ValueError(f"Cache file has embedding dtype mismatch: {m.embed_dtype} vs {embedder.embed_dtype}")
		if self.cache_size != m.total_size:
			raise ValueError(f"Cache file has an unexpected actual size: {self.cache_size} vs {m.total_size}")
		if self.use_targets:
			if h.target_dim < 1 or h.embed_targets_dim < 1:
				raise ValueError("Cache file must have positive target dimensions")
			if m.target_dtype != embedder.token_dtype:
				raise ValueError(f"Cache file has target token IDs dtype mismatch: {m.target_dtype} vs {embedder.token_dtype}")
			if h.target_nouns_num - 1 > torch.iinfo(m.embed_targets_dtype).max:
				raise ValueError("Cache file embedding target noun IDs dtype is not big enough for the number of target nouns")
		self.enter_count = 0
		self._file = self._mmap = None
		self.target_token_ids = self.target_mask = self.embed_targets = self.embed_target_weights = self.embeds = None

	def __len__(self) -> int:
		return self.header.embed_num

	def _array(self, dtype, count, offset, shape):
		return np.frombuffer(self._mmap, dtype=_NP[dtype], count=count, offset=offset).reshape(shape)

	def __enter__(self) -> "EmbeddingCache":
		h, m = self.header, self.meta
		if self.use_targets:
			tc = self.embedder.target_config
			if tc is None:
				raise ValueError("Cannot enter embedding cache that uses targets without a target configuration")
			if h.target_dim != tc.token_length:
				raise ValueError(f"Cache file has target token IDs dimension mismatch: {h.target_dim} vs {tc.token_length}")
			if m.target_dtype != tc.token_dtype or m.target_mask_dtype != tc.mask_dtype:
				raise ValueError("Cache file has target token dtype mismatch")
			if self.strict_embedder and h.embedder_strict:  # fixed_token_length never affects a cache; a cache written with masks serves a mask-free config too (reference :577-583)
				ex = {"fixed_token_length"}
				hashes = (self.embedder.get_configuration_hash(main_config=False, target_config=True, target_exclude=ex, hexdigest=False),
				          self.embedder.get_configuration_hash(main_config=False, target_config=True, target_exclude=ex, target_override={"use_masks": True}, hexdigest=False))
				if h.target_config_hash not in hashes:
					raise ValueError("Cache file target config hash does not match target config hash => Incompatible")
		if self._mmap is None:
			self._file = open(self.cache_path, "rb")
			st = os.fstat(self._file.fileno())
			if (st.st_ino, st.st_dev, st.st_size, st.st_mtime_ns) != (self.cache_stat.st_ino, self.cache_stat.st_dev, self.cache_stat.st_size, self.cache_stat.st_mtime_ns):
				self._file.close()
				raise ValueError("Cache file has externally changed since it was first opened")
			self._mmap = mmap.mmap(self._file.fileno(), length=0, access=mmap.ACCESS_READ)
			if self._mmap[:Header.STRUCT.size] != self.header_bytes:
				raise ValueError("Cache file header bytes have changed since cache file was first opened")
			R, C, N, M, F = h.target_nouns_num, h.target_dim, h.embed_num, h.embed_targets_dim, h.embed_dim
			if self.use_targets:
				self.target_token_ids = self._array(m.target_dtype, R * C, m.target_offset, (R, C))
				self.target_mask = self._array(m.target_mask_dtype, R * C, m.target_mask_offset, (R, C)) if self.embedder.target_config.use_masks else None
				self.embed_targets = self._array(m.embed_targets_dtype, N * M, m.embed_targets_offset, (N, M))
				self.embed_target_weights = self._array(m.embed_dtype, N * M, m.embed_target_weights_offset, (N, M))
			self.embeds = self._array(m.embed_dtype, N * F, m.embed_offset, (N, F))
		self.enter_count += 1
		return self

	def __exit__(self, exc_type, exc_val, exc_tb) -> bool:
		self.enter_count -= 1
		if self.enter_count <= 0:
			self.enter_count = 0
			self.target_token_ids = self.target_mask = self.embed_targets = self.embed_target_weights = self.embeds = None
			if self._mmap is not None:
				try:
					self._mmap.close()
				except BufferError:
					pass  # numpy views still alive somewhere: the map is released when they are
				self._file.close()
				self._mmap = self._file = None
		return False

	def get_samples(self, start: int, stop: int, use_weights: bool = True):
		"""-> embed B x F, target_ids B x M, target B x M x C, mask B x M x C | None, weight B x M | None (host tensors; reference :690-723)."""
		if self._mmap is None:
			raise RuntimeError("Cache must be entered before data can be accessed")
		if start < 0 or stop < 0:
			raise IndexError("Negative indices are not supported")
		stop = min(stop, self.header.embed_num)
		t = lambda a: torch.from_numpy(np.array(a))  # copies: the mmap is read-only
		embed = t(self.embeds[start:stop]) if stop > start else torch.empty((0, self.header.embed_dim), dtype=self.meta.embed_dtype)
		if not self.use_targets:
			return embed, None, None, None, None
		ids = self.embed_targets[start:stop]
		target = t(self.target_token_ids[ids])
		mask = None if self.target_mask is None else t(self.target_mask[ids])
		weight = t(self.embed_target_weights[start:stop]) if use_weights else None
		return embed, t(ids), target, mask, weight

	def create_dataset(self, batch_size: int, training: bool) -> "CacheDataset":
		return CacheDataset(self, batch_size, training)


class CacheDataset:
	"""Batch-granular dataset over a cache (reference EmbeddingCache.Dataset :758-914): item i = batch i of the (rotated) epoch."""

	def __init__(self, embed_cache: EmbeddingCache, batch_size: int, training: bool):
		self.embed_cache, self.header, self.batch_size, self.training = embed_cache, embed_cache.header, batch_size, training
		h = self.header
		if batch_size < 1:
			raise ValueError(f"Batch size must be a positive integer: {batch_size}")
		if batch_size > h.embed_num:
			raise ValueError(f"Batch size cannot be larger than the number of embeddings in the cache: {batch_size} > {h.embed_num}")
		complete, leftover = divmod(h.embed_num, batch_size)
		self.num_embeds = h.embed_num - (leftover if training else 0)
		self.num_items = complete + (0 if training or leftover == 0 else 1)
		self.epoch_index_offset = 0
		self.embedder = embed_cache.embedder
		self.targets = embed_cache.target_nouns
		self.num_invalid_targets = 1 if embed_cache.target_nouns else 0
		self.num_valid_targets = len(self.targets) - self.num_invalid_targets if self.targets else 0
		self.use_targets = embed_cache.use_targets
		self.nominal_data_config = embedding_dataset.DataConfig(use_weights=not (h.default_weights and h.full_targets), unit_weights=h.unit_weights, multi_target=h.embed_targets_dim > 1,
		                                                        multi_first=False, full_targets=h.full_targets, fixed_multi_length=False, multi_length=h.embed_targets_dim or 1)
		self.data_config = self.nominal_data_config
		self.loader_info = embedding_dataset.LoaderInfo(num_workers=0, prefetch_factor=0, pin_memory=False, on_device=True, batch_size=batch_size,
		                                                batch_size_last=0 if training else leftover, complete_batches=complete, incomplete_batch=(not training and leftover > 0),
		                                                epoch_batches=self.num_items, epoch_samples=self.num_embeds, available_samples=self.num_embeds)

	def resolve_data_config(self, **data_kwargs) -> embedding_dataset.DataConfig:
		"""What the caller wants (None = don't care) merged with what the cache offers (reference embedding_dataset.py:122-150)."""
		nominal = dataclasses.asdict(self.nominal_data_config)
		merged = {k: (data_kwargs.pop(k) if data_kwargs.get(k) is not None else (data_kwargs.pop(k, None), v)[1]) for k, v in nominal.items()}
		if data_kwargs:
			raise ValueError(f"Cannot resolve invalid data config fields: {sorted(data_kwargs)}")
		dc = embedding_dataset.DataConfig.create(merged, use_targets=self.use_targets)
		if dc.multi_length > self.nominal_data_config.multi_length:
			raise ValueError(f"This embedding dataset does not support a number of multi-targets above {self.nominal_data_config.multi_length}: {dc.multi_length}")
		if not self.header.full_targets and merged["full_targets"] != nominal["full_targets"] and dc.full_targets != nominal["full_targets"]:
			raise ValueError("Incompatibility between embedding dataset and requested data config in fields: ['full_targets']")
		return dc

	def configure_data(self, data_config: embedding_dataset.DataConfig):
		"""reference embedding_dataset.py:152-159"""
		if self.use_targets and not self.embedder.target_config.use_masks and not data_config.use_weights and not data_config.full_targets:
			raise RuntimeError("When using non-full targets without padding masks and without weights, there is no robust way of being able to tell which targets are supposed to be ignored")
		self.data_config = data_config

	def loaded(self):
		return self.embed_cache

	def __len__(self) -> int:
		return self.num_items

	def batch_range(self, index: int) -> tuple[int, int]:
		"""(start, count) of batch `index`; rows are (start + i) % N.  Reference :832-843."""
		if index < 0 or index >= self.num_items:
			raise IndexError("Index out of range")
		N, B = self.header.embed_num, self.batch_size
		if self.epoch_index_offset == 0 or not self.training:
			start = index * B
			return start, min(B, N - start)
		return (index * B + self.epoch_index_offset) % N, B

	def __getitem__(self, index: int):
		"""Host-side batch, bit-identical to the reference's Dataset.__getitem__ (used as the checker of the device loader)."""
		start, count = self.batch_range(index)
		N = self.header.embed_num
		uw = self.data_config.use_weights
		if start + count <= N:
			embed, ids, target, mask, weight = self.embed_cache.get_samples(start, start + count, use_weights=uw)
		else:
			a = self.embed_cache.get_samples(start, N, use_weights=uw)
			b = self.embed_cache.get_samples(0, start + count - N, use_weights=uw)
			embed, ids, target, mask, weight = (None if x is None else torch.cat((x, y), dim=0) for x, y in zip(a, b))
		if ids is None:
			return embed, None, None, None
		target, mask, weight = finish_batch(self.data_config, self.header, self.embedder.target_config, ids, target, mask, weight)
		return embed, target, mask, weight


def finish_batch(dc: embedding_dataset.DataConfig, h: Header, tc: embedders.TargetConfig, ids, target, mask, weight):
	"""Trimming / weight rules of reference :845-895 on host tensors."""
	if dc.multi_target:
		trimmed = dc.multi_length < target.shape[1]
		if trimmed:
			target = target[:, :dc.multi_length]
			mask = None if mask is None else mask[:, :dc.multi_length]
			if weight is None:
				ids = ids[:, :dc.multi_length]
			else:
				weight = weight[:, :dc.multi_length]
		if not dc.fixed_multi_length and target.shape[1] > 1:
			present = (ids if weight is None else weight).bool().any(dim=0)
			if not bool(present.all()):
				keep = int(present.int().argmin())
				target = target[:, :keep]
				mask = None if mask is None else mask[:, :keep]
				weight = None if weight is None else weight[:, :keep]
		if weight is not None and dc.unit_weights and (not h.unit_weights or trimmed):
			weight = torch.ones_like(weight) if weight.shape[1] == 1 else torch.nn.functional.normalize(weight, p=1, dim=1)
	else:
		target = target[:, 0]
		mask = None if mask is None else mask[:, 0]
		if weight is not None:
			m = weight.shape[1]
			weight = weight[:, 0]
			if dc.unit_weights and (not h.unit_weights or m > 1):
				weight = torch.ones_like(weight)
	if not tc.fixed_token_length and mask is not None:
		col_all = (mask.all(dim=0) if mask.ndim > 2 else mask).all(dim=0)
		if bool(col_all.any()):
			keep = int(col_all.int().argmax())
			target, mask = target[..., :keep], mask[..., :keep]
	if dc.multi_target and dc.multi_first:
		target = target.transpose(0, 1)
		mask = None if mask is None else mask.transpose(0, 1)
		weight = None if weight is None else weight.transpose(0, 1)
	return target, mask, weight


class GroupSlice(tuple):
	"""One loader batch (embed, target, mask, weight) that is rows [pos * B, (pos + 1) * B) of a group's buffers `full`: the micro-batches of an optimizer step assembled by one
	launch.  A plain tuple to every consumer; train.train_step takes `full` when it is handed all `size` slices of one group in order."""

	def __new__(cls, part, full, pos, size):
		self = super().__new__(cls, part)
		self.full, self.pos, self.size = full, pos, size
		return self


class DeviceLoader:
	"""HBM-resident cache + on-device batch assembly.  Iterating yields (embed, target, mask, weight) device tensors with exactly the values
	``CacheDataset.__getitem__`` produces for the same epoch offset and batch order; `rank`/`world` stride the batch sequence for data parallel
	training: rank r takes batches r, r + world, ... of the shared shuffled order, and every rank takes the SAME number of them (the
	num_items % world batches at the end of the order are left out of a training epoch -- ranks that ran different numbers of optimizer steps would
	pair up mismatched gradient all-reduces).  The order and the epoch rotation come from `seed`, which therefore must be given, and equal, on every
	rank when world > 1."""

	def __init__(self, dataset: CacheDataset, device: torch.device, *, seed: Optional[int] = None, rank: int = 0, world: int = 1, hbm_budget_bytes: Optional[int] = None,
	             stream_depth: int = 4, group: int = 1):
		self.group = max(1, min(32, int(group)))  # loader batches assembled by ONE launch (the micro-batches of an optimizer step: action_train passes its accumulation factor)
		if hbm_budget_bytes is None:  # (200 GB of the 288: what a cache may take before its vectors stream; $NOVIC_LOADER_HBM_BUDGET overrides -- bench.py forces streaming with 0)
			hbm_budget_bytes = int(os.environ.get("NOVIC_LOADER_HBM_BUDGET", 200 << 30))
		self.ds, self.device, self.rank, self.world = dataset, device, rank, world
		if world < 1 or not 0 <= rank < world:
			raise ValueError(f"Bad data-parallel coordinates: rank {rank} of {world}")
		if world > 1 and seed is None:
			raise ValueError("A data-parallel loader needs an explicit seed (the same on every rank): the ranks stride ONE shuffled batch order")
		if world > 1 and dataset.training and dataset.num_items < world:
			raise ValueError(f"Fewer batches ({dataset.num_items}) than ranks ({world})")
		self.rng = random.Random(seed)
		cache, h = dataset.embed_cache, dataset.header
		if cache.meta.embed_dtype != torch.float32:
			raise NotImplementedError("device loader handles float32 embedding caches")
		# A cache beyond the HBM budget keeps its embedding vectors (the bulk: N x F floats) in the memory-mapped file and STREAMS the rows of the next
		# `stream_depth` batches -- contiguous slices of the file -- through pinned staging buffers and a copy stream into device slabs; the token table
		# and the per-embedding target ids / weights (a few bytes per embedding) stay resident either way.
		self.streaming = cache.meta.total_size > hbm_budget_bytes
		self._open = None
		with cache:
			up = lambda a: torch.from_numpy(np.array(a)).to(device)
			self.embeds = None if self.streaming else up(cache.embeds)
			self.use_targets = cache.use_targets
			if self.use_targets:
				self.tok = up(cache.target_token_ids)
				self.msk = None if cache.target_mask is None else up(cache.target_mask).view(torch.uint8)
				self.ids = up(cache.embed_targets.astype(np.int32))
				self.wts = up(cache.embed_target_weights)
				# host-side per-noun label length (first all-padding column) and per-embedding target ids: the trims are decided without touching the device
				mask_h = np.array(cache.target_mask) if cache.target_mask is not None else None
				self.noun_len = None if mask_h is None else np.where(mask_h.all(axis=1), 0, mask_h.shape[1] - mask_h[:, ::-1].argmin(axis=1)).astype(np.int64)
				self.ids_h = np.array(cache.embed_targets)
				self.wts_h = np.array(cache.embed_target_weights)

		if self.streaming:
			D, B, F = max(2, int(stream_depth)), dataset.batch_size, h.embed_dim
			self._pinned = [torch.empty((B, F), dtype=torch.float32).pin_memory() for _ in range(D)]
			self._slabs = [torch.empty((B, F), dtype=torch.float32, device=device) for _ in range(D)]
			self._copied = [torch.cuda.Event() for _ in range(D)]   # slab d holds its batch (copy stream)
			self._consumed = [None] * D                              # the gather that read slab d has run (compute stream)
			self._copy_stream = ops.named_stream(device, "loader")

	def __len__(self) -> int:
		n = self.ds.num_items
		if self.ds.training:
			return n // self.world  # equal on every rank
		return (n - self.rank + self.world - 1) // self.world  # evaluation: every batch is visited once, no collective depends on the count

	@property
	def loader_info(self) -> embedding_dataset.LoaderInfo:
		"""This rank's share of the dataset's LoaderInfo (what GradAccum is built from under data parallelism)."""
		li = self.ds.loader_info
		if self.world == 1:
			return li
		n = len(self)
		last_is_mine = li.incomplete_batch and (self.ds.num_items - 1) % self.world == self.rank
		complete = n - (1 if last_is_mine else 0)
		samples = complete * li.batch_size + (li.batch_size_last if last_is_mine else 0)
		return dataclasses.replace(li, complete_batches=complete, incomplete_batch=last_is_mine, batch_size_last=li.batch_size_last if last_is_mine else 0, epoch_batches=n,
		                           epoch_samples=samples, available_samples=samples)

	def state_dict(self) -> dict:
		"""The shuffle / rotation generator (checkpointed so that a resumed run continues the batch sequence instead of replaying epoch 1's)."""
		return dict(rng=self.rng.getstate())

	def load_state_dict(self, state: dict):
		self.rng.setstate(state["rng"])

	def __iter__(self) -> Iterator:
		ds = self.ds
		ds.epoch_index_offset = self.rng.randrange(ds.num_embeds) if ds.training else 0
		order = list(range(ds.num_items))
		if ds.training:
			self.rng.shuffle(order)
		mine = order[self.rank::self.world][:len(self)]
		if not self.streaming:
			G = self.group
			if G > 1:
				for k in range(0, len(mine), G):
					chunk = mine[k:k + G]
					if len(chunk) == G:
						yield from self.assemble_group(chunk)
					else:
						for index in chunk:
							yield self.assemble(index)
				return
			for index in mine:
				yield self.assemble(index)
			return
		D = len(self._slabs)
		with self.ds.embed_cache as cache:  # the memory map stays open for the epoch
			self._open = cache
			try:
				nfull = 0
				if self.group > 1:  # whole groups first: staged by a thread of their own, one gather launch each (_iter_streamed_groups); the remainder batch by batch below
					nfull = len(mine) // self.group * self.group
					yield from self._iter_streamed_groups(mine[:nfull])
					mine = mine[nfull:]
				for k in range(min(D - 1, len(mine))):
					self._stage(mine[k], k % D)
				for k, index in enumerate(mine):
					if k + D - 1 < len(mine):
						self._stage(mine[k + D - 1], (k + D - 1) % D)  # the slot whose batch was consumed one iteration ago
					torch.cuda.current_stream(self.device).wait_event(self._copied[k % D])
					out = self.assemble(index, slot=k % D)
					ev = torch.cuda.Event()
					ev.record(torch.cuda.current_stream(self.device))
					self._consumed[k % D] = ev
					yield out
			finally:
				self._open = None

	def _iter_streamed_groups(self, mine):
		"""Streaming, `group` batches at a time: a staging THREAD copies each group's rows file -> pinned step buffer -> device step slab (copy stream) up to two groups ahead
		-- the host-side copy out of the page cache (1 MB per batch, ~100 us each) was most of the loader's host time, and numpy releases the interpreter lock for it --
		while this thread turns a staged slab into batches with one gather launch (assemble_group).  Ring of three step slabs; a slab is refilled only behind the gather that
		read it (event), a pinned buffer only behind the copy out of it."""
		import queue
		import threading
		G, B, F, dev = self.group, self.ds.batch_size, self.ds.header.embed_dim, self.device
		S = 3
		if getattr(self, "_gslabs", None) is None:
			self._gpinned = [torch.empty((G * B, F), dtype=torch.float32).pin_memory() for _ in range(S)]
			self._gslabs = [torch.empty((G * B, F), dtype=torch.float32, device=dev) for _ in range(S)]
			self._gcopied = [torch.cuda.Event() for _ in range(S)]
			self._gconsumed = [None] * S  # the gather that read slab s has run (compute stream)
		# The consumption events live on the LOADER, next to the slabs they guard, and survive the iteration: the host runs several steps ahead of the device and
		# training_loop synchronises per chunk, not per epoch, so the gathers of one epoch's last groups can still be queued when the next epoch (or a new iteration
		# after an early close) stages its first groups into the same slabs -- a per-iteration list would let those copies overtake them.
		consumed = self._gconsumed
		groups = [mine[k:k + G] for k in range(0, len(mine), G)]
		free = threading.Semaphore(S)      # slabs this thread has released (their gather is enqueued, its event recorded)
		ready = queue.Queue()              # (group number | exception) in order
		stop = threading.Event()
		N, src = self.ds.header.embed_num, self._open.embeds
		dev_index = dev.index if dev.index is not None else torch.cuda.current_device()

		def stage():
			try:
				torch.cuda.set_device(dev_index)
				for k, chunk in enumerate(groups):
					free.acquire()
					if stop.is_set():
						return
					slot = k % S
					self._gcopied[slot].synchronize()  # the previous copy OUT of this pinned buffer is done (no-op for a never-recorded event)
					dst = self._gpinned[slot].numpy()
					for g, index in enumerate(chunk):
						start, count = self.ds.batch_range(index)
						first = min(count, N - start % N)
						dst[g * B:g * B + first] = src[start % N:start % N + first]
						if first < count:
							dst[g * B + first:g * B + count] = src[:count - first]  # wrap
					if consumed[slot] is not None:
						self._copy_stream.wait_event(consumed[slot])  # the gather that read this slab has run
					with torch.cuda.stream(self._copy_stream):
						self._gslabs[slot].copy_(self._gpinned[slot], non_blocking=True)
						self._gcopied[slot].record(self._copy_stream)
					ready.put(k)
			except BaseException as e:  # (handed to the consumer: a dead staging thread must not leave it waiting)
				ready.put(e)

		t = threading.Thread(target=stage, name="novic-loader-stage", daemon=True)
		t.start()
		try:
			for k, chunk in enumerate(groups):
				got = ready.get()
				if isinstance(got, BaseException):
					raise got
				slot = k % S
				torch.cuda.current_stream(dev).wait_event(self._gcopied[slot])
				out = list(self.assemble_group(chunk, slab=self._gslabs[slot]))
				ev = torch.cuda.Event()
				ev.record(torch.cuda.current_stream(dev))
				consumed[slot] = ev
				free.release()
				yield from out
		finally:
			stop.set()
			for _ in range(S):
				free.release()
			t.join()

	def _stage(self, index: int, slot: int):
		"""Rows [start, start + count) mod N of the file -> pinned buffer -> device slab `slot` (copy stream)."""
		start, count = self.ds.batch_range(index)
		N = self.ds.header.embed_num
		self._copied[slot].synchronize()  # the previous copy OUT of this pinned buffer is done (no-op for a never-recorded event)
		dst = self._pinned[slot].numpy()
		src = self._open.embeds
		first = min(count, N - start % N)
		dst[:first] = src[start % N:start % N + first]
		if first < count:
			dst[first:count] = src[:count - first]  # wrap
		if self._consumed[slot] is not None:
			self._copy_stream.wait_event(self._consumed[slot])  # the gather that read this slab has run
		with torch.cuda.stream(self._copy_stream):
			self._slabs[slot][:count].copy_(self._pinned[slot][:count], non_blocking=True)
			self._copied[slot].record(self._copy_stream)

	def _group_shape(self, ranges):
		"""(M targets, C tokens) of every batch of the group as assemble() would choose them -- absent trailing targets and the token length are trimmed per BATCH, by the
		batch's content -- if all the batches agree, else None.  One vectorised pass over the group's rows instead of one per batch."""
		ds, h, dc, tc = self.ds, self.ds.header, self.ds.data_config, self.ds.embedder.target_config
		M_file, C = h.embed_targets_dim, h.target_dim
		M = min(dc.multi_length, M_file) if dc.multi_target else 1
		dyn_m = dc.multi_target and not dc.fixed_multi_length and M > 1
		dyn_c = not tc.fixed_token_length and self.noun_len is not None
		if not (dyn_m or dyn_c):
			return M, C
		G, count = len(ranges), ranges[0][1]
		rows = (np.asarray([r[0] for r in ranges], dtype=np.int64)[:, None] + np.arange(count, dtype=np.int64)[None, :]) % h.embed_num  # [G][count]
		ids = self.ids_h[rows, :M]  # [G][count][M]
		if dyn_m:
			present = ((self.wts_h[rows, :M] != 0) if dc.use_weights else (ids != 0)).any(axis=1)  # [G][M]
			ms = np.where(present.all(axis=1), M, present.argmin(axis=1))
			if (ms != ms[0]).any():
				return None
			M = int(ms[0])
			ids = ids[:, :, :M]
		if dyn_c:
			if ids.size == 0:
				return M, 0
			cs = self.noun_len[ids].reshape(G, -1).max(axis=1)
			if (cs != cs[0]).any():
				return None
			C = int(cs[0])
		return M, C

	def assemble_group(self, indices, slab: Optional[torch.Tensor] = None):
		"""len(indices) full batches by ONE launch into one set of buffers (novic_cache_gather_group), handed out as GroupSlice tuples -- views of the group's buffers, each
		exactly what assemble(index) returns -- which train_step recognises and uses whole instead of concatenating its micro-batches.  Batches whose trimmed target shapes
		differ are assembled one by one.  slab (streaming loader): a staged [G * B][F] buffer that already holds the group's embedding rows, batch after batch."""
		ds, h, dc = self.ds, self.ds.header, self.ds.data_config
		G = len(indices)
		ranges = [ds.batch_range(i) for i in indices]
		count = ranges[0][1]
		shape = self._group_shape(ranges) if self.use_targets else (0, 0)
		if shape is None or any(r[1] != count for r in ranges):
			for g, index in enumerate(indices):
				yield self.assemble(index) if slab is None else self.assemble(index, table=slab, table_row0=g * ds.batch_size)
			return
		M, C = shape
		starts = [r[0] for r in ranges]
		N, F = h.embed_num, h.embed_dim
		embed = torch.empty((G * count, F), dtype=torch.float32, device=self.device)
		table, staged = (self.embeds, -1) if slab is None else (slab, 0)
		if not self.use_targets:
			ops.cache_gather_group(table, None, None, None, None, starts, count, N, F, 0, 0, 0, 0, embed, None, None, None, 0, staged_row0=staged)
			full = (embed, None, None, None)
		else:
			M_file, C_file = h.embed_targets_dim, h.target_dim
			trimmed = dc.multi_target and dc.multi_length < M_file
			wmode = 0  # (as assemble)
			if dc.use_weights and dc.unit_weights:
				if dc.multi_target:
					if not h.unit_weights or trimmed:
						wmode = 2 if M == 1 else 1
				elif not h.unit_weights or M_file > 1:
					wmode = 2
			target = torch.empty((G * count, M, C), dtype=self.tok.dtype, device=self.device)
			mask = torch.empty((G * count, M, C), dtype=torch.uint8, device=self.device) if self.msk is not None else None
			weight = torch.empty((G * count, M), dtype=torch.float32, device=self.device) if dc.use_weights else None
			ops.cache_gather_group(table, self.ids, self.tok, self.msk, self.wts, starts, count, N, F, M_file, C_file, M, C, embed, target, mask, weight, wmode, staged_row0=staged)
			if not dc.multi_target:
				target = target[:, 0]
				mask = None if mask is None else mask[:, 0]
				weight = None if weight is None else weight[:, 0]
			mask = None if mask is None else mask.view(torch.bool)
			full = (embed, target, mask, weight)
		multi_first = self.use_targets and dc.multi_target and dc.multi_first
		for g in range(G):
			part = tuple(None if t is None else t[g * count:(g + 1) * count] for t in full)
			if multi_first:  # (M first, as assemble hands it out; such batches are not stacked along the sample dimension: train._mergeable)
				part = (part[0],) + tuple(None if t is None else t.transpose(0, 1) for t in part[1:])
			yield GroupSlice(part, full, g, G)

	def assemble(self, index: int, slot: Optional[int] = None, table: Optional[torch.Tensor] = None, table_row0: int = 0):
		ds, h, dc, tc = self.ds, self.ds.header, self.ds.data_config, self.ds.embedder.target_config
		start, count = ds.batch_range(index)
		N, F = h.embed_num, h.embed_dim
		embed = torch.empty((count, F), dtype=torch.float32, device=self.device)
		if table is not None:  # (a staged buffer holding the batch's rows from row table_row0 on: the grouped streaming loader's step slab)
			staged = int(table_row0)
		else:
			table, staged = (self.embeds, -1) if slot is None else (self._slabs[slot], 0)
		if not self.use_targets:
			ops.cache_gather(table, None, None, None, None, start, count, N, F, 0, 0, 0, 0, embed, None, None, None, 0, staged_row0=staged)
			return embed, None, None, None
		rows = (start + np.arange(count)) % N
		M_file, C_file = h.embed_targets_dim, h.target_dim
		M = min(dc.multi_length, M_file) if dc.multi_target else 1
		trimmed = dc.multi_target and dc.multi_length < M_file
		ids = self.ids_h[rows, :M]
		if dc.multi_target and not dc.fixed_multi_length and M > 1:
			present = ((self.wts_h[rows, :M] != 0) if dc.use_weights else (ids != 0)).any(axis=0)
			if not present.all():
				M = int(present.argmin())
				ids = ids[:, :M]
		C = C_file
		if not tc.fixed_token_length and self.noun_len is not None:
			C = int(self.noun_len[ids].max()) if ids.size else 0
		wmode = 0  # 0 copy, 1 L1-normalise over the kept targets, 2 ones (reference :866-870, :881-882)
		if dc.use_weights and dc.unit_weights:
			if dc.multi_target:
				if not h.unit_weights or trimmed:
					wmode = 2 if M == 1 else 1
			elif not h.unit_weights or M_file > 1:
				wmode = 2
		target = torch.empty((count, M, C), dtype=self.tok.dtype, device=self.device)
		mask = torch.empty((count, M, C), dtype=torch.uint8, device=self.device) if self.msk is not None else None
		weight = torch.empty((count, M), dtype=torch.float32, device=self.device) if dc.use_weights else None
		ops.cache_gather(table, self.ids, self.tok, self.msk, self.wts, start, count, N, F, M_file, C_file, M, C, embed, target, mask, weight, wmode, staged_row0=staged)
		if not dc.multi_target:
			target = target[:, 0]
			mask = None if mask is None else mask[:, 0]
			weight = None if weight is None else weight[:, 0]
		elif dc.multi_first:
			target = target.transpose(0, 1)
			mask = None if mask is None else mask.transpose(0, 1)
			weight = None if weight is None else weight.transpose(0, 1)
		return embed, target, (None if mask is None else mask.view(torch.bool)), weight
