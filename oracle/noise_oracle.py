"""CPU oracle: train-time embedding noise (TEST INFRASTRUCTURE ONLY).

Restates embedding_noise.py:72-75 (GaussElem), :78-95 (GaussVec), :105-112 (angle rotation),
:131-132 / :151-152 (angle draws), :169-172 (mixture) and the optional mean shift
(train.py:1263-1265) as pure functions of *explicit* random tensors, so the HIP kernel's
"injected randoms" mode is comparable value-for-value and its Philox mode statistically.
Never imported by novic_amd/.
"""
from __future__ import annotations

import math

import torch

_normalize = lambda x: torch.nn.functional.normalize(x, dim=-1)  # eps 1e-12, as F.normalize


def mean_shift(embed: torch.Tensor, shift: torch.Tensor) -> torch.Tensor:
	return _normalize(embed + shift)


def gauss_elem(embed: torch.Tensor, z: torch.Tensor, vec_norm: float) -> torch.Tensor:
	"""z ~ N(0, I) of embed's shape; elem std = vec_norm / sqrt(F)."""
	return _normalize(embed + z * (vec_norm / math.sqrt(embed.shape[-1])))


def gauss_vec(embed: torch.Tensor, z: torch.Tensor, r: torch.Tensor, vec_norm: float) -> torch.Tensor:
	"""z ~ N(0, I) direction, r ~ N(0, 1) per row (B x 1)."""
	return _normalize(embed + _normalize(z) * r * vec_norm)


def rotate(embed: torch.Tensor, z: torch.Tensor, angle: torch.Tensor) -> torch.Tensor:
	"""Rotate each unit row by `angle` (B x 1, radians) towards the component of z orthogonal to it."""
	dirn = z - embed * (embed * z).sum(dim=1, keepdim=True)
	dirn = _normalize(dirn)
	return _normalize(embed * angle.cos() + dirn * angle.sin())


def gauss_angle_draw(r: torch.Tensor, angle_std_deg: float, angle_max_deg: float) -> torch.Tensor:
	a = math.radians(angle_max_deg)
	return (r * math.radians(angle_std_deg)).clamp(min=-a, max=a)


def uniform_angle_draw(u: torch.Tensor, angle_min_deg: float, angle_max_deg: float) -> torch.Tensor:
	"""u ~ U[0, 1) per row -> angle in [min, max) radians."""
	lo, hi = math.radians(angle_min_deg), math.radians(angle_max_deg)
	return lo + u * (hi - lo)


def gauss_elem_uniform_angle(embed: torch.Tensor, z_gauss: torch.Tensor, z_angle: torch.Tensor, u_angle: torch.Tensor, u_mix: torch.Tensor,
                             vec_norm: float, angle_min_deg: float, angle_max_deg: float, mix_ratio: float) -> torch.Tensor:
	"""Per row: with probability mix_ratio the rotated row, else the Gaussian-perturbed row."""
	rotated = rotate(embed, z_angle, uniform_angle_draw(u_angle, angle_min_deg, angle_max_deg))
	perturbed = gauss_elem(embed, z_gauss, vec_norm)
	return torch.where(u_mix < mix_ratio, rotated, perturbed)
