"""Train-time embedding noise behind the reference's ``EmbeddingNoise`` surface (reference embedding_noise.py:15-173).

Every scheme is ONE launch of the fused HIP kernel ``novic_noise_fused`` (mean shift, Gaussian add / rotation / mixture and the
final L2 renorm in registers, Philox in-kernel) instead of the reference's ~12 elementwise + 4 RNG kernels.
``forward(embed)`` modifies ``embed`` in place and returns it, as the reference documents (:49-52).
"""
from __future__ import annotations

import math
from typing import Optional

import torch

from . import ops


class EmbeddingNoise(torch.nn.Module):

	@staticmethod
	def create(scheme: str, embed_dim: int, vec_norm: float, angle_min: float, angle_max: float, angle_std: float, mix_ratio: float) -> Optional["EmbeddingNoise"]:
		if not scheme:
			return None
		key = scheme.lower()
		if key == "gausselem":
			return GaussElemNoise(embed_dim=embed_dim, vec_norm=vec_norm)
		if key == "gaussvec":
			return GaussVecNoise(embed_dim=embed_dim, vec_norm=vec_norm)
		if key == "gaussangle":
			return GaussAngleNoise(embed_dim=embed_dim, angle_std=angle_std, angle_max=angle_max)
		if key == "uniformangle":
			return UniformAngleNoise(embed_dim=embed_dim, angle_min=angle_min, angle_max=angle_max)
		if key == "gausselemuniformangle":
			return GaussElemUniformAngleNoise(embed_dim=embed_dim, vec_norm=vec_norm, angle_min=angle_min, angle_max=angle_max, mix_ratio=mix_ratio)
		raise ValueError(f"Unsupported embedding noise type: {scheme}")

	mode = ops.NOISE_NONE

	def __init__(self, scheme: str, embed_dim: int, seed: int = 0x5EED):
		super().__init__()
		self.scheme, self.embed_dim = scheme, embed_dim
		self.seed = seed
		self.calls = 0  # Philox offset: a fresh stream per call
		self.mean_shift: Optional[torch.Tensor] = None  # optional F vector fused in front of the noise (train.py:1263-1265)

	def params(self) -> dict:
		return {}

	def forward(self, embed: torch.Tensor, *, inj_z1=None, inj_z2=None, inj_row=None, inj_mix=None) -> torch.Tensor:
		if embed.ndim != 2 or embed.shape[1] != self.embed_dim or embed.dtype != torch.float32:
			raise ValueError(f"Expected a Bx{self.embed_dim} float32 tensor of unit rows")
		self.calls += 1
		return ops.noise_fused(embed, self.mode, seed=self.seed, offset=self.calls, inj_z1=inj_z1, inj_z2=inj_z2, inj_row=inj_row, inj_mix=inj_mix,
		                       mean_shift=self.mean_shift, **self.params())


class MeanShiftOnly(EmbeddingNoise):
	"""embed <- normalize(embed + mean_shift) with no noise (train.py:1263-1265 when noise_scheme is empty)."""

	def __init__(self, embed_dim: int, mean_shift: torch.Tensor):
		super().__init__(scheme="", embed_dim=embed_dim)
		self.mean_shift = mean_shift.reshape(-1).contiguous()


class GaussElemNoise(EmbeddingNoise):
	mode = ops.NOISE_GAUSS_ELEM

	def __init__(self, embed_dim: int, vec_norm: float):
		super().__init__(scheme="GaussElem", embed_dim=embed_dim)
		self.vec_norm = vec_norm
		self.elem_std = vec_norm / math.sqrt(embed_dim)
		if self.elem_std <= 0:
			raise ValueError(f"Element noise standard deviation must be positive: {self.elem_std:.3g}")

	def extra_repr(self) -> str:
		return f"embed_dim={self.embed_dim}, vec_norm={self.vec_norm:.3g}, elem_std={self.elem_std:.3g}"

	def params(self):
		return dict(vec_norm=self.vec_norm)


class GaussVecNoise(EmbeddingNoise):
	mode = ops.NOISE_GAUSS_VEC

	def __init__(self, embed_dim: int, vec_norm: float):
		super().__init__(scheme="GaussVec", embed_dim=embed_dim)
		self.vec_norm = vec_norm
		if vec_norm <= 0:
			raise ValueError(f"Vector noise norm must be positive: {vec_norm:.3g}")

	def extra_repr(self) -> str:
		return f"embed_dim={self.embed_dim}, vec_norm={self.vec_norm:.3g}"

	def params(self):
		return dict(vec_norm=self.vec_norm)


class AngleNoise(EmbeddingNoise):
	pass


class GaussAngleNoise(AngleNoise):
	mode = ops.NOISE_GAUSS_ANGLE

	def __init__(self, embed_dim: int, angle_std: float, angle_max: float):
		super().__init__(scheme="GaussAngle", embed_dim=embed_dim)
		self.angle_std, self.angle_max = angle_std, angle_max
		self.angle_std_rad, self.angle_max_rad = math.radians(angle_std), math.radians(angle_max)
		if self.angle_std_rad <= 0 or self.angle_max_rad <= 0:
			raise ValueError(f"Angular noise standard deviation and maximum value must both be positive: std {self.angle_std_rad:.3g} radians, max {self.angle_max_rad:.3g} radians")

	def extra_repr(self) -> str:
		return f"embed_dim={self.embed_dim}, angle_std={self.angle_std:.3g}\xB0, angle_max={self.angle_max:.3g}\xB0"

	def params(self):
		return dict(angle_std=self.angle_std_rad, angle_max=self.angle_max_rad)


class UniformAngleNoise(AngleNoise):
	mode = ops.NOISE_UNIFORM_ANGLE

	def __init__(self, embed_dim: int, angle_min: float, angle_max: float):
		super().__init__(scheme="UniformAngle", embed_dim=embed_dim)
		self.angle_min, self.angle_max = angle_min, angle_max
		self.angle_min_rad, self.angle_max_rad = math.radians(angle_min), math.radians(angle_max)
		if self.angle_min_rad > self.angle_max_rad:
			raise ValueError(f"Minimum angular noise must be smaller than maximum angular noise: min {self.angle_min_rad:.3g} radians, max {self.angle_max_rad:.3g} radians")

	def extra_repr(self) -> str:
		return f"embed_dim={self.embed_dim}, angle_min={self.angle_min:.3g}\xB0, angle_max={self.angle_max:.3g}\xB0"

	def params(self):
		return dict(angle_min=self.angle_min_rad, angle_max=self.angle_max_rad)


class GaussElemUniformAngleNoise(EmbeddingNoise):
	mode = ops.NOISE_GAUSS_ELEM_UNIFORM_ANGLE

	def __init__(self, embed_dim: int, vec_norm: float, angle_min: float, angle_max: float, mix_ratio: float):
		super().__init__(scheme="GaussElemUniformAngle", embed_dim=embed_dim)
		self.gauss_elem_noise = GaussElemNoise(embed_dim=embed_dim, vec_norm=vec_norm)
		self.uniform_angle_noise = UniformAngleNoise(embed_dim=embed_dim, angle_min=angle_min, angle_max=angle_max)
		self.mix_ratio = mix_ratio
		if mix_ratio < 0 or mix_ratio > 1:
			raise ValueError(f"Mix ratio must be in the range [0, 1]: {mix_ratio:.3g}")

	def extra_repr(self) -> str:
		return f"mix_ratio={self.mix_ratio:.3g}"

	def params(self):
		return dict(vec_norm=self.gauss_elem_noise.vec_norm, angle_min=self.uniform_angle_noise.angle_min_rad, angle_max=self.uniform_angle_noise.angle_max_rad,
		            mix_ratio=self.mix_ratio)
