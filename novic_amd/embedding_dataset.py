"""Data contracts consumed by the decoder and the training loop (host side).

Mirrors reference embedding_dataset.py: ``DataConfig`` (:19-42), ``LoaderInfo`` (:45-57), ``GradAccum`` (:198-273) -- same field
names, same arithmetic -- so configs/checkpoints written by either side read the same.  Pure host logic, no kernels.
"""
from __future__ import annotations

import dataclasses
import itertools
from typing import Iterable, Optional, Union

import torch


@dataclasses.dataclass(frozen=True)
class DataConfig:
	use_weights: bool
	unit_weights: bool
	multi_target: bool
	multi_first: bool
	full_targets: bool
	fixed_multi_length: bool
	multi_length: int

	@staticmethod
	def create(data_config_dict: dict[str, Union[bool, int]], use_targets: bool = True) -> "DataConfig":
		d = dict(data_config_dict)
		if not use_targets:
			d.update(use_weights=False, multi_target=False)
		if not d["use_weights"]:
			d.update(unit_weights=True)
		if not d["multi_target"]:
			d.update(multi_first=False, full_targets=True, fixed_multi_length=True, multi_length=1)
		cfg = DataConfig(**d)
		if cfg.multi_length < 1:
			raise ValueError(f"Number of multi-targets needs to be positive: {cfg.multi_length}")
		return cfg

	@staticmethod
	def single() -> "DataConfig":
		return DataConfig.create(dict(use_weights=False, unit_weights=True, multi_target=False, multi_first=False, full_targets=True, fixed_multi_length=True, multi_length=1))


@dataclasses.dataclass(frozen=True)
class LoaderInfo:
	num_workers: int
	prefetch_factor: int
	pin_memory: bool
	on_device: bool
	batch_size: int
	batch_size_last: int
	complete_batches: int
	incomplete_batch: bool
	epoch_batches: int
	epoch_samples: int
	available_samples: int


class GradAccum:
	"""Gradient-accumulation bookkeeping: which loader batches form an optimizer step and how each mean batch loss is scaled."""

	def __init__(self, loader, loader_info: LoaderInfo, accum_size: int, drop_last: bool):
		if accum_size < 1:
			raise ValueError(f"Accumulation size must be at least 1: {accum_size}")
		assert loader_info.epoch_batches == len(loader)
		self.raw_loader, self.raw_loader_info = loader, loader_info
		self.accum_size, self.drop_last = accum_size, drop_last
		self.batch_size = loader_info.batch_size
		self.accum_batch_size = self.batch_size * accum_size
		self.complete_steps = loader_info.complete_batches // accum_size
		self.complete_batches = self.complete_steps * accum_size
		self.complete_samples = self.complete_batches * self.batch_size
		if drop_last:
			self.loader_batches, self.loader_samples = self.complete_batches, self.complete_samples
			self.incomplete_batches = self.incomplete_samples = 0
			self.incomplete_step = False
		else:
			self.loader_batches, self.loader_samples = loader_info.epoch_batches, loader_info.epoch_samples
			self.incomplete_batches = self.loader_batches - self.complete_batches
			self.incomplete_samples = self.loader_samples - self.complete_samples
			assert self.incomplete_batches >= 0 and self.incomplete_samples >= 0 and (self.incomplete_batches > 0) == (self.incomplete_samples > 0)
			self.incomplete_step = self.incomplete_samples > 0
		self.loader_steps = self.complete_steps + self.incomplete_step
		self.batch_num = 0

	def loader(self) -> Iterable[tuple[torch.Tensor, Optional[torch.Tensor], Optional[torch.Tensor], Optional[torch.Tensor]]]:
		self.batch_num = 0
		if self.drop_last and self.loader_batches < self.raw_loader_info.epoch_batches:
			return itertools.islice(self.raw_loader, self.loader_batches)
		return self.raw_loader

	def loss_scale(self, num_in_batch: int) -> tuple[float, bool]:
		"""Host-side twin of accum_loss: (scale applied to this batch's mean loss, whether an optimizer step follows it)."""
		self.batch_num += 1
		scale = 1.0 / self.accum_size if self.batch_num <= self.complete_batches else num_in_batch / self.incomplete_samples
		step = self.batch_num % self.accum_size == 0 or self.batch_num == self.raw_loader_info.epoch_batches
		return scale, step

	def accum_loss(self, mean_batch_loss: torch.Tensor, num_in_batch: int) -> tuple[torch.Tensor, bool]:
		scale, step = self.loss_scale(num_in_batch)
		return mean_batch_loss * scale, step
