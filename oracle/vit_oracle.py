"""CPU oracle: CLIP ViT image tower (TEST INFRASTRUCTURE ONLY).

The arithmetic behind the reference's ``inference_image`` (embedders.py:589-594, :759-764, :902-907) lives in third-party packages that
are NOT vendored in the reference tree: ``open_clip_torch==2.23`` (requirements.txt:8), ``git+https://github.com/openai/CLIP.git``
(unpinned, requirements.txt:4) and ``transformers`` (README pin 4.38.2).  This file restates their published vision transformer
(conv1 stride=patch without bias -> [class; patches] + positional embedding -> ln_pre -> pre-LN blocks with biased in_proj/out_proj and
a 4x MLP with GELU (OpenCLIP) or QuickGELU (OpenAI) -> ln_post on the class token -> projection without bias) followed by the reference's
own fp32 cast + F.normalize.  It is pinned against ``transformers.CLIPVisionModelWithProjection`` built from an explicit local config
(tests/golden/make_golden_vit.py); real-weight parity is unpinned (weights are unreachable offline).
State-dict keys follow OpenCLIP's ``visual.*`` naming.  Never imported by novic_amd/.
"""
from __future__ import annotations

import dataclasses
import math

import torch


@dataclasses.dataclass(frozen=True)
class ViTSpec:
	image_size: int = 224
	patch_size: int = 32
	width: int = 768
	layers: int = 12
	heads: int = 12
	mlp_ratio: float = 4.0
	embed_dim: int = 512
	quick_gelu: bool = False
	ln_eps: float = 1e-5

	@property
	def grid(self):
		return self.image_size // self.patch_size

	@property
	def tokens(self):
		return self.grid ** 2 + 1

	@property
	def mlp_dim(self):
		return int(self.width * self.mlp_ratio)


def init_state_dict(spec: ViTSpec, seed: int = 0) -> dict[str, torch.Tensor]:
	"""Seeded random weights with CLIP-like scales (enough spread for a meaningful numerics check)."""
	g = torch.Generator().manual_seed(seed)
	W, L, F, M = spec.width, spec.layers, spec.embed_dim, spec.mlp_dim
	n = lambda *shape, std: torch.randn(*shape, generator=g) * std
	sc = W ** -0.5
	sd = {
		"visual.conv1.weight": n(W, 3, spec.patch_size, spec.patch_size, std=0.02),
		"visual.class_embedding": n(W, std=sc),
		"visual.positional_embedding": n(spec.tokens, W, std=sc),
		"visual.ln_pre.weight": 1 + n(W, std=0.05), "visual.ln_pre.bias": n(W, std=0.05),
		"visual.ln_post.weight": 1 + n(W, std=0.05), "visual.ln_post.bias": n(W, std=0.05),
		"visual.proj": n(W, F, std=sc),
	}
	for i in range(L):
		p = f"visual.transformer.resblocks.{i}."
		sd[p + "ln_1.weight"] = 1 + n(W, std=0.05); sd[p + "ln_1.bias"] = n(W, std=0.05)
		sd[p + "ln_2.weight"] = 1 + n(W, std=0.05); sd[p + "ln_2.bias"] = n(W, std=0.05)
		sd[p + "attn.in_proj_weight"] = n(3 * W, W, std=sc); sd[p + "attn.in_proj_bias"] = n(3 * W, std=0.02)
		sd[p + "attn.out_proj.weight"] = n(W, W, std=sc * (2 * L) ** -0.5); sd[p + "attn.out_proj.bias"] = n(W, std=0.02)
		sd[p + "mlp.c_fc.weight"] = n(M, W, std=(2 * W) ** -0.5); sd[p + "mlp.c_fc.bias"] = n(M, std=0.02)
		sd[p + "mlp.c_proj.weight"] = n(W, M, std=sc * (2 * L) ** -0.5); sd[p + "mlp.c_proj.bias"] = n(W, std=0.02)
	return sd


def _r(x, bf16):
	return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def _lin(x, w, b, bf16):
	y = _r(x, bf16) @ _r(w, bf16).T
	return y if b is None else y + b


def _ln(x, w, b, eps):
	return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def _h(x):
	return x.to(torch.float16).to(torch.float32)


def _lin_half(x, w, b):
	"""A half-precision linear: fp16 operands, fp32 accumulation, the result (+ the half bias) rounded to half."""
	y = _h(x) @ _h(w).T
	return _h(y if b is None else y + _h(b))


def encode_image_half(sd: dict, spec: ViTSpec, images: torch.Tensor, normalize: bool = True) -> torch.Tensor:
	"""clip's HALF-PRECISION model, which the reference runs for its 'openai:' embedders (embedders.py:488-489: `manual_amp_dtype = torch.float16`, the model as
	`clip.load` returns it on a GPU: `convert_weights` -> fp16 weights; clip/model.py: `LayerNorm.forward` casts to fp32, normalises, casts back; everything else in half):
	every tensor between two operations is rounded to IEEE half, matrix products accumulate in fp32, the residual stream is half, LayerNorm statistics are fp32.
	Pinned against transformers' CLIP vision tower run in torch.float16 on the CPU (tests/golden/make_golden_vit_half.py).  The reference's own epilogue is the same as for
	the other towers: `.to(float32)`, `F.normalize` (embedders.py:593-594)."""
	B = images.shape[0]
	W, H = spec.width, spec.heads
	D = W // H
	p = spec.patch_size
	ln = lambda x, w, b: _h(_ln(x, _h(w), _h(b), spec.ln_eps))
	patches = _h(images).unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5).reshape(B, spec.grid ** 2, 3 * p * p)
	x = _lin_half(patches, sd["visual.conv1.weight"].reshape(W, -1), None)
	x = _h(torch.cat((_h(sd["visual.class_embedding"]).expand(B, 1, W), x), dim=1) + _h(sd["visual.positional_embedding"]))
	x = ln(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])
	N = x.shape[1]
	for i in range(spec.layers):
		q = f"visual.transformer.resblocks.{i}."
		h = ln(x, sd[q + "ln_1.weight"], sd[q + "ln_1.bias"])
		qkv = _lin_half(h, sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"]).view(B, N, 3, H, D)
		qq, kk, vv = (qkv[:, :, c].transpose(1, 2) for c in range(3))
		att = _h(torch.softmax(_h(_h(qq * (1.0 / math.sqrt(D))) @ kk.transpose(-1, -2)), dim=-1))
		o = _h(att @ vv).transpose(1, 2).reshape(B, N, W)
		x = _h(x + _lin_half(o, sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"]))
		h = ln(x, sd[q + "ln_2.weight"], sd[q + "ln_2.bias"])
		h = _lin_half(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"])
		h = _h(h * _h(torch.sigmoid(_h(1.702 * h)))) if spec.quick_gelu else _h(torch.nn.functional.gelu(h))
		x = _h(x + _lin_half(h, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"]))
	cls = ln(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])
	out = _h(cls @ _h(sd["visual.proj"]))
	return torch.nn.functional.normalize(out.float(), dim=-1) if normalize else out.float()


def encode_image(sd: dict, spec: ViTSpec, images: torch.Tensor, bf16: bool = False, normalize: bool = True, half_stream: bool = False) -> torch.Tensor:
	"""images B x 3 x R x R f32 -> B x F f32 (unit rows when normalize).  bf16=True rounds GEMM operands/outputs like the HIP kernels.
	half_stream (with bf16): the rounding points of the HIP tower in its half-stream mode (NativeViT.half_stream, round 6) -- bf16 GEMM operands as ever, but the residual
	stream is IEEE half: ln_pre's output, and out = half(x + half(linear)) at the two residual adds (novic_amd/csrc/gemm_epilogue.hpp: resid_f16_elem)."""
	hs = (lambda t: t.to(torch.float16).to(torch.float32)) if half_stream else (lambda t: t)
	B = images.shape[0]
	W, H = spec.width, spec.heads
	D = W // H
	p = spec.patch_size
	patches = images.unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5).reshape(B, spec.grid ** 2, 3 * p * p)
	x = _r(_lin(patches, sd["visual.conv1.weight"].reshape(W, -1), None, bf16), bf16)
	x = torch.cat((sd["visual.class_embedding"].expand(B, 1, W), x), dim=1) + sd["visual.positional_embedding"]
	x = hs(_ln(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], spec.ln_eps))
	N = x.shape[1]
	for i in range(spec.layers):
		q = f"visual.transformer.resblocks.{i}."
		h = _ln(x, sd[q + "ln_1.weight"], sd[q + "ln_1.bias"], spec.ln_eps)
		qkv = _r(_lin(h, sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"], bf16), bf16).view(B, N, 3, H, D)
		qq, kk, vv = (qkv[:, :, c].transpose(1, 2) for c in range(3))
		att = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(D), dim=-1)
		o = _r((_r(att, bf16) @ vv), bf16).transpose(1, 2).reshape(B, N, W)
		lin = _lin(o, sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"], bf16)
		x = hs(x + hs(lin)) if half_stream else x + _r(lin, bf16)
		h = _ln(x, sd[q + "ln_2.weight"], sd[q + "ln_2.bias"], spec.ln_eps)
		h = _lin(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"], bf16)
		h = _r(h * torch.sigmoid(1.702 * h) if spec.quick_gelu else torch.nn.functional.gelu(h), bf16)
		lin = _lin(h, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"], bf16)
		x = hs(x + hs(lin)) if half_stream else x + _r(lin, bf16)
	cls = _ln(x[:, 0], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], spec.ln_eps)
	out = _r(cls, bf16) @ _r(sd["visual.proj"], bf16)
	return torch.nn.functional.normalize(out.float(), dim=-1) if normalize else out.float()
