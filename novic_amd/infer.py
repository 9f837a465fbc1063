"""Inference-side host API (work in progress this round): generation configs and tasks behind the reference's infer.py surface."""
from __future__ import annotations

import dataclasses
import itertools
import re
from typing import Any, Optional, Sequence

import torch


def format_semifix(value: float, precision: int = 3) -> str:
	"""Compact number formatting used in generation-config names (reference utils.format_semifix): fixed notation, trailing zeros stripped."""
	s = f"{value:.{precision}f}".rstrip("0").rstrip(".")
	return s if s not in ("", "-0") else "0"


@dataclasses.dataclass(frozen=True)
class GenerationConfig:
	"""reference infer.py:357-433"""
	method: str
	topk: int
	vocab_prior: bool
	vocab_per_token: bool
	vocab_scaler: float
	guided: bool
	guide_renorm: bool
	temperature: float
	length_alpha: float
	name: str = dataclasses.field(init=False)

	def __post_init__(self):
		object.__setattr__(self, "name", self.generate_name())

	def generate_name(self) -> str:
		prior = f"{'tok' if self.vocab_per_token else 'tgt'}{format_semifix(self.vocab_scaler)}" if self.vocab_prior else "none"
		guide = "n" if not self.guided else ("r" if self.guide_renorm else "p")
		return f"{self.method}_k{self.topk}_v{prior}_g{guide}_t{format_semifix(self.temperature)}_a{format_semifix(self.length_alpha)}"

	@staticmethod
	def from_name(name: str) -> "GenerationConfig":
		parts = name.split("_")
		kw = dict(method=parts[0], topk=0, vocab_prior=False, vocab_per_token=False, vocab_scaler=0.0, guided=False, guide_renorm=False, temperature=1.0, length_alpha=0.0)
		for part in parts[1:]:
			if not part:
				raise ValueError(f"Unexpected multiple underscores in generation configuration: {name}")
			key, val = part[:1], part[1:]
			try:
				if key == "k":
					kw["topk"] = int(val)
				elif key == "v":
					if val != "none":
						m = re.fullmatch(r"(tok|tgt)(.*)", val)
						if not m:
							raise ValueError(f"Invalid vocab prior specification: {val}")
						kw.update(vocab_prior=True, vocab_per_token=(m.group(1) == "tok"), vocab_scaler=float(m.group(2)))
				elif key == "g":
					if val not in ("n", "p", "r"):
						raise ValueError(f"Invalid guide specification: {val}")
					kw.update(guided=(val != "n"), guide_renorm=(val == "r"))
				elif key == "t":
					kw["temperature"] = float(val)
				elif key == "a":
					kw["length_alpha"] = float(val)
				else:
					raise ValueError(f"Invalid prefix: {key}")
			except ValueError:
				raise ValueError(f"Failed to parse generation configuration part: {part}")
		cfg = GenerationConfig(**kw)
		if cfg.method not in ("greedy", "beam", "all"):
			raise ValueError(f"Invalid generation configuration method: {cfg.method}")
		if cfg.topk < 1:
			raise ValueError(f"Missing or invalid non-positive generation configuration top-k: {cfg.topk}")
		if cfg.temperature <= 0:
			raise ValueError(f"Invalid non-positive generation configuration temperature tau: {cfg.temperature}")
		assert cfg.name == name
		return cfg
