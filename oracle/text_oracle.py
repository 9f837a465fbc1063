"""CPU oracle: CLIP text tower (TEST INFRASTRUCTURE ONLY).

The arithmetic behind the reference's ``inference_tokens`` / ``inference_text`` (embedders.py:423-426, :557-583, :728-753) lives in third-party
packages that are NOT vendored in the reference tree: ``open_clip_torch==2.23`` (requirements.txt:8), ``git+https://github.com/openai/CLIP.git``
(unpinned) and ``transformers`` (README pin 4.38.2).  This file restates their published text transformer: token embedding + positional
embedding -> pre-LN blocks with a CAUSAL attention mask (biased in_proj / out_proj, 4x MLP with GELU (OpenCLIP) or QuickGELU (OpenAI)) -> ln_final
-> the row of the END-OF-TEXT token (= arg-max token id, the largest id of the CLIP vocabulary) -> text_projection without bias, followed by the
reference's own fp32 cast + F.normalize (:583).  Pinned against ``transformers.CLIPTextModelWithProjection`` built from an explicit local config
(tests/golden/make_golden_text.py); real-weight parity is unpinned (weights and BPE vocabulary are unreachable offline).
State-dict keys follow OpenCLIP naming.  Never imported by novic_amd/.
"""
from __future__ import annotations

import dataclasses
import math

import torch


@dataclasses.dataclass(frozen=True)
class TextSpec:
	vocab_size: int = 49408
	context_length: int = 77
	width: int = 512
	layers: int = 12
	heads: int = 8
	mlp_ratio: float = 4.0
	embed_dim: int = 512
	quick_gelu: bool = False
	ln_eps: float = 1e-5

	@property
	def mlp_dim(self):
		return int(self.width * self.mlp_ratio)


def init_state_dict(spec: TextSpec, seed: int = 0) -> dict[str, torch.Tensor]:
	g = torch.Generator().manual_seed(seed)
	W, L, F, M = spec.width, spec.layers, spec.embed_dim, spec.mlp_dim
	n = lambda *shape, std: torch.randn(*shape, generator=g) * std
	sc = W ** -0.5
	sd = {
		"token_embedding.weight": n(spec.vocab_size, W, std=0.02),
		"positional_embedding": n(spec.context_length, W, std=0.01),
		"ln_final.weight": 1 + n(W, std=0.05), "ln_final.bias": n(W, std=0.05),
		"text_projection": n(W, F, std=sc),
	}
	for i in range(L):
		p = f"transformer.resblocks.{i}."
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


def encode_text_half(sd: dict, spec: TextSpec, token_ids: torch.Tensor, normalize: bool = True, eot_token_id=None) -> torch.Tensor:
	"""clip's HALF-PRECISION text tower -- what the reference runs for 'openai:' embedders (embedders.py:488-489, :582-583; clip/model.py: fp16 weights and activations,
	LayerNorm computed in fp32 and cast back, the projection as a half matmul): every tensor between two operations rounded to IEEE half, matrix products accumulated
	in fp32.  Pinned against transformers' CLIP text tower run in torch.float16 on the CPU (tests/golden/make_golden_text_half.py)."""
	B, S = token_ids.shape
	W, H = spec.width, spec.heads
	D = W // H
	ln = lambda t, w, b: _h(_ln(t, _h(w), _h(b), spec.ln_eps))
	x = _h(_h(sd["token_embedding.weight"])[token_ids.long()] + _h(sd["positional_embedding"][:S]))
	causal = torch.full((S, S), float("-inf")).triu(1)
	for i in range(spec.layers):
		q = f"transformer.resblocks.{i}."
		h = ln(x, sd[q + "ln_1.weight"], sd[q + "ln_1.bias"])
		qkv = _lin_half(h, sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"]).view(B, S, 3, H, D)
		qq, kk, vv = (qkv[:, :, c].transpose(1, 2) for c in range(3))
		att = _h(torch.softmax(_h(_h(qq * (1.0 / math.sqrt(D))) @ kk.transpose(-1, -2)) + causal, dim=-1))
		o = _h(att @ vv).transpose(1, 2).reshape(B, S, W)
		x = _h(x + _lin_half(o, sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"]))
		h = ln(x, sd[q + "ln_2.weight"], sd[q + "ln_2.bias"])
		h = _lin_half(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"])
		h = _h(h * _h(torch.sigmoid(_h(1.702 * h)))) if spec.quick_gelu else _h(torch.nn.functional.gelu(h))
		x = _h(x + _lin_half(h, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"]))
	pos = token_ids.long().argmax(dim=1) if eot_token_id is None else (token_ids == eot_token_id).int().argmax(dim=1)
	pooled = ln(x, sd["ln_final.weight"], sd["ln_final.bias"])[torch.arange(B), pos]  # (clip normalises every position, then picks the END-OF-TEXT row: the same row either way)
	out = _h(pooled @ _h(sd["text_projection"]))
	return torch.nn.functional.normalize(out.float(), dim=-1) if normalize else out.float()


def encode_text(sd: dict, spec: TextSpec, token_ids: torch.Tensor, bf16: bool = False, normalize: bool = True, eot_token_id=None, half_stream: bool = False) -> torch.Tensor:
	"""token_ids B x S (S <= context_length) integer -> B x F f32 (unit rows when normalize).  bf16=True rounds GEMM operands/outputs like the HIP kernels.
	half_stream (with bf16): the rounding points of the HIP tower in its half-stream mode (NativeTextTower.half_stream, round 6): bf16 GEMM operands as ever, the residual
	stream IEEE half -- the embedding sum, and out = half(x + half(linear)) at the two residual adds."""
	hs = _h if half_stream else (lambda t: t)
	B, S = token_ids.shape
	W, H = spec.width, spec.heads
	D = W // H
	x = hs(sd["token_embedding.weight"][token_ids.long()] + sd["positional_embedding"][:S])
	causal = torch.full((S, S), float("-inf")).triu(1)
	for i in range(spec.layers):
		q = f"transformer.resblocks.{i}."
		h = _ln(x, sd[q + "ln_1.weight"], sd[q + "ln_1.bias"], spec.ln_eps)
		qkv = _r(_lin(h, sd[q + "attn.in_proj_weight"], sd[q + "attn.in_proj_bias"], bf16), bf16).view(B, S, 3, H, D)
		qq, kk, vv = (qkv[:, :, c].transpose(1, 2) for c in range(3))
		att = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(D) + causal, dim=-1)
		o = _r((_r(att, bf16) @ vv), bf16).transpose(1, 2).reshape(B, S, W)
		lin = _lin(o, sd[q + "attn.out_proj.weight"], sd[q + "attn.out_proj.bias"], bf16)
		x = hs(x + hs(lin)) if half_stream else x + _r(lin, bf16)
		h = _ln(x, sd[q + "ln_2.weight"], sd[q + "ln_2.bias"], spec.ln_eps)
		h = _lin(h, sd[q + "mlp.c_fc.weight"], sd[q + "mlp.c_fc.bias"], bf16)
		h = _r(h * torch.sigmoid(1.702 * h) if spec.quick_gelu else torch.nn.functional.gelu(h), bf16)
		lin = _lin(h, sd[q + "mlp.c_proj.weight"], sd[q + "mlp.c_proj.bias"], bf16)
		x = hs(x + hs(lin)) if half_stream else x + _r(lin, bf16)
	# CLIP vocabulary: the END-OF-TEXT token has the largest id -> arg-max (OpenAI CLIP / open_clip / HF with eos_token_id == 2); any other
	# vocabulary: first occurrence of the given end id (HF's pooling for eos_token_id != 2)
	pos = token_ids.long().argmax(dim=1) if eot_token_id is None else (token_ids == eot_token_id).int().argmax(dim=1)
	pooled = x[torch.arange(B), pos]
	pooled = _ln(pooled, sd["ln_final.weight"], sd["ln_final.bias"], spec.ln_eps)
	out = _r(pooled, bf16) @ _r(sd["text_projection"], bf16)
	return torch.nn.functional.normalize(out.float(), dim=-1) if normalize else out.float()
