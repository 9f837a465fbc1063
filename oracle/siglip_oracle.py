"""CPU oracle: SigLIP image and text towers as open_clip builds them (TEST INFRASTRUCTURE ONLY).

Two of the four released NOVIC checkpoints name SigLIP embedders (`openclip:timm/ViT-B-16-SigLIP`, `openclip:timm/ViT-SO400M-14-SigLIP`: reference README.md:293-298),
which the reference reaches through `open_clip.create_model_and_transforms('hf-hub:' + id)` (embedders.py:680-689): a timm vision trunk (`open_clip.timm_model.TimmModel`:
`vit_*_siglip_*`, no class token, global_pool = 'map') and open_clip's own `TextTransformer` with `no_causal_mask`, `pool_type = 'last'`, `proj_bias` -- third-party code that
is not in the reference tree and not installed here.  This file restates the published architecture:

  image: conv patch embedding WITH bias -> + learned positions (no class token, no pre-norm) -> pre-LN blocks (eps 1e-6, biased qkv / proj, 4x MLP, exact-erf GELU in the timm
         release the reference's environment resolves to) -> final LayerNorm over every token -> attention pooling (timm `AttentionPoolLatent`: one learned latent query,
         q / kv / proj linears, softmax(q k^T / sqrt(d)) v over all tokens, then x + mlp(norm(x))) -> the pooled vector IS the embedding (timm_proj 'none');
  text:  token + positional embedding -> pre-LN blocks WITHOUT a causal mask -> final LayerNorm -> the LAST position -> Linear projection with bias.

State-dict keys are open_clip's (`visual.trunk.*` = timm's names, `text.*`).  Pinned against `transformers.SiglipModel` (hidden_act='gelu') built from an explicit local
config through the key map of tests/golden/make_golden_siglip.py; real-weight parity is unpinned (weights unreachable offline).  Never imported by novic_amd/.
"""
from __future__ import annotations

import dataclasses
import math

import torch


@dataclasses.dataclass(frozen=True)
class SigLIPVisionSpec:
	image_size: int = 224
	patch_size: int = 16
	width: int = 768
	layers: int = 12
	heads: int = 12
	mlp_dim: int = 3072
	ln_eps: float = 1e-6
	gelu_tanh: bool = False  # act_kwargs.approximate = 'tanh' (= transformers' hidden_act 'gelu_pytorch_tanh')

	@property
	def tokens(self):
		return (self.image_size // self.patch_size) ** 2


@dataclasses.dataclass(frozen=True)
class SigLIPTextSpec:
	vocab_size: int = 32000
	context_length: int = 64
	width: int = 768
	layers: int = 12
	heads: int = 12
	mlp_dim: int = 3072
	embed_dim: int = 768
	ln_eps: float = 1e-6
	gelu_tanh: bool = False


def init_vision_state_dict(spec: SigLIPVisionSpec, seed: int = 0) -> dict:
	g = torch.Generator().manual_seed(seed)
	W, L, M, p = spec.width, spec.layers, spec.mlp_dim, spec.patch_size
	n = lambda *shape, std: torch.randn(*shape, generator=g) * std
	sc = W ** -0.5
	t = "visual.trunk."
	sd = {t + "patch_embed.proj.weight": n(W, 3, p, p, std=0.02), t + "patch_embed.proj.bias": n(W, std=0.02), t + "pos_embed": n(1, spec.tokens, W, std=sc),
	      t + "norm.weight": 1 + n(W, std=0.05), t + "norm.bias": n(W, std=0.05),
	      t + "attn_pool.latent": n(1, 1, W, std=sc), t + "attn_pool.q.weight": n(W, W, std=sc), t + "attn_pool.q.bias": n(W, std=0.02),
	      t + "attn_pool.kv.weight": n(2 * W, W, std=sc), t + "attn_pool.kv.bias": n(2 * W, std=0.02), t + "attn_pool.proj.weight": n(W, W, std=sc),
	      t + "attn_pool.proj.bias": n(W, std=0.02), t + "attn_pool.norm.weight": 1 + n(W, std=0.05), t + "attn_pool.norm.bias": n(W, std=0.05),
	      t + "attn_pool.mlp.fc1.weight": n(M, W, std=(2 * W) ** -0.5), t + "attn_pool.mlp.fc1.bias": n(M, std=0.02),
	      t + "attn_pool.mlp.fc2.weight": n(W, M, std=sc * 0.5), t + "attn_pool.mlp.fc2.bias": n(W, std=0.02)}
	for i in range(L):
		b = f"{t}blocks.{i}."
		sd[b + "norm1.weight"] = 1 + n(W, std=0.05); sd[b + "norm1.bias"] = n(W, std=0.05)
		sd[b + "norm2.weight"] = 1 + n(W, std=0.05); sd[b + "norm2.bias"] = n(W, std=0.05)
		sd[b + "attn.qkv.weight"] = n(3 * W, W, std=sc); sd[b + "attn.qkv.bias"] = n(3 * W, std=0.02)
		sd[b + "attn.proj.weight"] = n(W, W, std=sc * (2 * L) ** -0.5); sd[b + "attn.proj.bias"] = n(W, std=0.02)
		sd[b + "mlp.fc1.weight"] = n(M, W, std=(2 * W) ** -0.5); sd[b + "mlp.fc1.bias"] = n(M, std=0.02)
		sd[b + "mlp.fc2.weight"] = n(W, M, std=sc * (2 * L) ** -0.5); sd[b + "mlp.fc2.bias"] = n(W, std=0.02)
	return sd


def init_text_state_dict(spec: SigLIPTextSpec, seed: int = 0) -> dict:
	g = torch.Generator().manual_seed(seed)
	W, L, M, F = spec.width, spec.layers, spec.mlp_dim, spec.embed_dim
	n = lambda *shape, std: torch.randn(*shape, generator=g) * std
	sc = W ** -0.5
	sd = {"text.token_embedding.weight": n(spec.vocab_size, W, std=0.02), "text.positional_embedding": n(spec.context_length, W, std=0.01),
	      "text.ln_final.weight": 1 + n(W, std=0.05), "text.ln_final.bias": n(W, std=0.05), "text.text_projection.weight": n(F, W, std=sc), "text.text_projection.bias": n(F, std=0.02)}
	for i in range(L):
		b = f"text.transformer.resblocks.{i}."
		sd[b + "ln_1.weight"] = 1 + n(W, std=0.05); sd[b + "ln_1.bias"] = n(W, std=0.05)
		sd[b + "ln_2.weight"] = 1 + n(W, std=0.05); sd[b + "ln_2.bias"] = n(W, std=0.05)
		sd[b + "attn.in_proj_weight"] = n(3 * W, W, std=sc); sd[b + "attn.in_proj_bias"] = n(3 * W, std=0.02)
		sd[b + "attn.out_proj.weight"] = n(W, W, std=sc * (2 * L) ** -0.5); sd[b + "attn.out_proj.bias"] = n(W, std=0.02)
		sd[b + "mlp.c_fc.weight"] = n(M, W, std=(2 * W) ** -0.5); sd[b + "mlp.c_fc.bias"] = n(M, std=0.02)
		sd[b + "mlp.c_proj.weight"] = n(W, M, std=sc * (2 * L) ** -0.5); sd[b + "mlp.c_proj.bias"] = n(W, std=0.02)
	return sd


def _r(x, bf16):
	return x.to(torch.bfloat16).to(torch.float32) if bf16 else x


def _lin(x, w, b, bf16):
	y = _r(x, bf16) @ _r(w, bf16).T
	return y if b is None else y + b


def _ln(x, w, b, eps):
	return torch.nn.functional.layer_norm(x, (x.shape[-1],), w, b, eps)


def _gelu(x, tanh):
	return torch.nn.functional.gelu(x, approximate="tanh" if tanh else "none")


def _block(x, sd, pre, names, H, eps, bf16, tanh=False):
	"""One pre-LN block without a mask.  names = (norm1, qkv weight, qkv bias, proj, norm2, fc1, fc2) key stems."""
	B, N, W = x.shape
	D = W // H
	n1, qw, qb, pj, n2, f1, f2 = names
	h = _ln(x, sd[pre + n1 + ".weight"], sd[pre + n1 + ".bias"], eps)
	qkv = _r(_lin(h, sd[pre + qw], sd[pre + qb], bf16), bf16).view(B, N, 3, H, D)
	qq, kk, vv = (qkv[:, :, c].transpose(1, 2) for c in range(3))
	att = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(D), dim=-1)
	o = _r((_r(att, bf16) @ vv), bf16).transpose(1, 2).reshape(B, N, W)
	x = x + _r(_lin(o, sd[pre + pj + ".weight"], sd[pre + pj + ".bias"], bf16), bf16)
	h = _ln(x, sd[pre + n2 + ".weight"], sd[pre + n2 + ".bias"], eps)
	h = _r(_gelu(_lin(h, sd[pre + f1 + ".weight"], sd[pre + f1 + ".bias"], bf16), tanh), bf16)
	return x + _r(_lin(h, sd[pre + f2 + ".weight"], sd[pre + f2 + ".bias"], bf16), bf16)


def encode_image(sd: dict, spec: SigLIPVisionSpec, images: torch.Tensor, bf16: bool = False, normalize: bool = True) -> torch.Tensor:
	"""images B x 3 x R x R f32 -> B x W f32 (unit rows when normalize).  bf16=True rounds GEMM operands / outputs like the HIP kernels."""
	B, W, H, p = images.shape[0], spec.width, spec.heads, spec.patch_size
	t = "visual.trunk."
	g = spec.image_size // p
	patches = images.unfold(2, p, p).unfold(3, p, p).permute(0, 2, 3, 1, 4, 5).reshape(B, g * g, 3 * p * p)
	x = _r(_lin(patches, sd[t + "patch_embed.proj.weight"].reshape(W, -1), sd[t + "patch_embed.proj.bias"], bf16), bf16) + sd[t + "pos_embed"]
	for i in range(spec.layers):
		x = _block(x, sd, f"{t}blocks.{i}.", ("norm1", "attn.qkv.weight", "attn.qkv.bias", "attn.proj", "norm2", "mlp.fc1", "mlp.fc2"), H, spec.ln_eps, bf16,
		           spec.gelu_tanh)
	x = _ln(x, sd[t + "norm.weight"], sd[t + "norm.bias"], spec.ln_eps)
	# attention pooling: one latent query over all tokens
	a = t + "attn_pool."
	D = W // H
	q = _r(_lin(sd[a + "latent"].view(1, W), sd[a + "q.weight"], sd[a + "q.bias"], bf16), bf16).view(1, 1, H, D).transpose(1, 2).expand(B, H, 1, D)
	kv = _r(_lin(x, sd[a + "kv.weight"], sd[a + "kv.bias"], bf16), bf16).view(B, -1, 2, H, D)
	kk, vv = kv[:, :, 0].transpose(1, 2), kv[:, :, 1].transpose(1, 2)
	att = torch.softmax(q @ kk.transpose(-1, -2) / math.sqrt(D), dim=-1)
	o = _r(_r(att, bf16) @ vv, bf16).transpose(1, 2).reshape(B, W)
	y = _lin(o, sd[a + "proj.weight"], sd[a + "proj.bias"], bf16)
	h = _ln(y, sd[a + "norm.weight"], sd[a + "norm.bias"], spec.ln_eps)
	h = _r(_gelu(_lin(h, sd[a + "mlp.fc1.weight"], sd[a + "mlp.fc1.bias"], bf16), spec.gelu_tanh), bf16)
	out = y + _r(_lin(h, sd[a + "mlp.fc2.weight"], sd[a + "mlp.fc2.bias"], bf16), bf16)
	return torch.nn.functional.normalize(out.float(), dim=-1) if normalize else out.float()


def encode_text(sd: dict, spec: SigLIPTextSpec, token_ids: torch.Tensor, bf16: bool = False, normalize: bool = True) -> torch.Tensor:
	"""token ids B x context_length (padded to the full context, as open_clip's tokenizer call does) -> B x F."""
	assert token_ids.shape[1] == spec.context_length
	x = sd["text.token_embedding.weight"][token_ids.long()] + sd["text.positional_embedding"]
	for i in range(spec.layers):
		x = _block(x, sd, f"text.transformer.resblocks.{i}.", ("ln_1", "attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj", "ln_2", "mlp.c_fc", "mlp.c_proj"), spec.heads,
		           spec.ln_eps, bf16, spec.gelu_tanh)
	pooled = _ln(x[:, -1], sd["text.ln_final.weight"], sd["text.ln_final.bias"], spec.ln_eps)
	out = _lin(pooled, sd["text.text_projection.weight"], sd["text.text_projection.bias"], bf16)
	return torch.nn.functional.normalize(out.float(), dim=-1) if normalize else out.float()
