"""SigLIP towers on the native kernels: what `open_clip.create_model_and_transforms('hf-hub:timm/ViT-B-16-SigLIP')` builds (reference embedders.py:680-689 for the released
checkpoints of README.md:293-298) -- a timm vision trunk without class token whose embedding is the attention-pooled token (`global_pool='map'`, `timm_proj='none'`), and
open_clip's text transformer without causal mask, pooled at the LAST position, projected by a Linear with bias.

Image launch sequence (all kernels, no torch arithmetic): im2col -> GEMM(patch embedding + bias, + positions as the fp32 residual operand) -> L x [LN -> GEMM qkv(+bias) ->
attention -> GEMM proj(+bias, +residual) -> LN -> GEMM fc1(+bias, +GELU) -> GEMM fc2(+bias, +residual)] -> LN over every token -> attention pooling: GEMM kv(+bias) into the
K / V columns of a qkv buffer whose Q columns hold the projected latent (a constant of the weights), the ordinary attention kernel, token 0 of every image -> GEMM proj(+bias)
-> LN -> GEMM fc1(+bias, +GELU) -> GEMM fc2(+bias, +residual) -> L2 normalise.  (The pooling runs the full N x N attention for the one row it needs: one layer's worth of
attention per forward, no new kernel.)

The text tower is `clip_text.NativeTextTower` with `causal=False, pool='last', proj_bias=True` (token rows are padded to the full context: without a causal mask the
padding takes part, as in open_clip, whose tokenizer call pads to `context_length`).

Weights use open_clip's own state-dict names (`visual.trunk.*` = timm's module names, `text.*`).  Dimensions are read off the tensors; the head count -- not visible in any
tensor -- comes from `vision_cfg.heads` when present, else from the timm model name, else width / 64.  Exact-erf GELU (what the timm release of the reference's environment
resolves to) unless the config asks for the tanh approximation (`act_kwargs.approximate = 'tanh'`, later open_clip / timm releases).

ViT-SO400M-14-SigLIP (width 1152, 16 heads of 72, MLP 4304; README.md:294) runs on the same kernels with ZERO-PADDED weights (clip_text.padded_head_dim): heads of 80 with the
soft-max scale of 72, an MLP of 4352 -- exact, the padding contributes zeros everywhere.
"""
from __future__ import annotations

import dataclasses
import re
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, clip_text, ops
from .clip_vit import make_image_transform
from .tower_runtime import TowerRuntime

# timm registry entries behind open_clip's SigLIP configs: name stem -> attention heads (width, depth, patch and MLP width are read off the tensors)
TIMM_HEADS = {"vit_base_patch16_siglip": 12, "vit_large_patch16_siglip": 16, "vit_so400m_patch14_siglip": 16}


@dataclasses.dataclass(frozen=True)
class SigLIPVisionConfig:
	image_size: int = 224
	patch_size: int = 16
	width: int = 768
	layers: int = 12
	heads: int = 12
	mlp_dim: int = 3072
	ln_eps: float = 1e-6
	gelu_tanh: bool = False  # vision_cfg.act_kwargs.approximate = 'tanh'

	@property
	def tokens(self) -> int:
		return (self.image_size // self.patch_size) ** 2

	@property
	def embed_dim(self) -> int:
		return self.width

	def flops_per_image(self) -> float:
		N, W, M = self.tokens, self.width, self.mlp_dim
		return self.layers * (N * (8 * W * W + 4 * W * M) + 4 * N * N * W) + 2 * 3 * self.patch_size ** 2 * W * N + N * 4 * W * W + 4 * N * N * W


def _pad8(n: int) -> int:
	return (n + 7) // 8 * 8


class NativeSigLIPViT(TowerRuntime, nn.Module):

	def __init__(self, cfg: SigLIPVisionConfig, seed: Optional[int] = None):
		super().__init__()
		self.cfg = cfg
		W, L, M, p = cfg.width, cfg.layers, cfg.mlp_dim, cfg.patch_size
		if W % cfg.heads or (W // cfg.heads) > clip_text.HEAD_DIMS[-1] or (W // cfg.heads) % 8 or W % 8 or M % 8:
			raise NotImplementedError(f"NativeSigLIPViT supports head_dim <= 80 in multiples of 8 (32 / 64 / 80 natively, others zero-padded; this model: {W} / {cfg.heads} = "
			                          f"{W / cfg.heads:g}) and widths that are multiples of 8")
		g = torch.Generator().manual_seed(seed) if seed is not None else None
		n = lambda *shape, std: nn.Parameter(torch.randn(*shape, generator=g) * std)
		sc = W ** -0.5
		self.names: list[str] = []

		def reg(name: str, param: nn.Parameter):
			self.names.append(name)
			self.register_parameter(name.replace(".", "__"), param)
		t = "visual.trunk."
		reg(t + "patch_embed.proj.weight", n(W, 3, p, p, std=0.02)); reg(t + "patch_embed.proj.bias", nn.Parameter(torch.zeros(W)))
		reg(t + "pos_embed", n(1, cfg.tokens, W, std=sc))
		for i in range(L):
			b = f"{t}blocks.{i}."
			for nm in ("norm1", "norm2"):
				reg(b + nm + ".weight", nn.Parameter(torch.ones(W))); reg(b + nm + ".bias", nn.Parameter(torch.zeros(W)))
			reg(b + "attn.qkv.weight", n(3 * W, W, std=sc)); reg(b + "attn.qkv.bias", nn.Parameter(torch.zeros(3 * W)))
			reg(b + "attn.proj.weight", n(W, W, std=sc * (2 * L) ** -0.5)); reg(b + "attn.proj.bias", nn.Parameter(torch.zeros(W)))
			reg(b + "mlp.fc1.weight", n(M, W, std=(2 * W) ** -0.5)); reg(b + "mlp.fc1.bias", nn.Parameter(torch.zeros(M)))
			reg(b + "mlp.fc2.weight", n(W, M, std=sc * (2 * L) ** -0.5)); reg(b + "mlp.fc2.bias", nn.Parameter(torch.zeros(W)))
		reg(t + "norm.weight", nn.Parameter(torch.ones(W))); reg(t + "norm.bias", nn.Parameter(torch.zeros(W)))
		a = t + "attn_pool."
		reg(a + "latent", n(1, 1, W, std=sc))
		reg(a + "q.weight", n(W, W, std=sc)); reg(a + "q.bias", nn.Parameter(torch.zeros(W)))
		reg(a + "kv.weight", n(2 * W, W, std=sc)); reg(a + "kv.bias", nn.Parameter(torch.zeros(2 * W)))
		reg(a + "proj.weight", n(W, W, std=sc)); reg(a + "proj.bias", nn.Parameter(torch.zeros(W)))
		reg(a + "norm.weight", nn.Parameter(torch.ones(W))); reg(a + "norm.bias", nn.Parameter(torch.zeros(W)))
		reg(a + "mlp.fc1.weight", n(M, W, std=(2 * W) ** -0.5)); reg(a + "mlp.fc1.bias", nn.Parameter(torch.zeros(M)))
		reg(a + "mlp.fc2.weight", n(W, M, std=sc * 0.5)); reg(a + "mlp.fc2.bias", nn.Parameter(torch.zeros(W)))
		for prm in self.parameters():
			prm.requires_grad_(False)
		self._w16: dict = {}
		self._w16_key = None
		self.preprocess: dict = {}

	def p(self, name: str) -> torch.Tensor:
		return getattr(self, name.replace(".", "__"))

	def state_dict(self, *args, **kwargs):
		return {n: self.p(n).detach() for n in self.names}

	def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
		missing = [n for n in self.names if n not in state_dict]
		extra = [k for k in state_dict if k not in self.names]
		if strict and (missing or extra):
			raise RuntimeError(f"NativeSigLIPViT.load_state_dict: missing {missing[:5]}, unexpected {extra[:5]}")
		with torch.no_grad():
			for n in self.names:
				if n in state_dict:
					self.p(n).copy_(state_dict[n].reshape(self.p(n).shape))
		self._w16_key = None

	def _shadow(self, device) -> dict:
		"""bf16 copies of the GEMM weights (patch embedding flattened, K padded to a multiple of 8) + the projected latent query, rebuilt when parameters change."""
		def ver(t):
			try:
				return t._version
			except RuntimeError:
				return 0
		key = (device, tuple(ver(self.p(n)) for n in self.names))
		if self._w16_key != key:
			cfg = self.cfg
			W = cfg.width
			K = 3 * cfg.patch_size ** 2
			conv = torch.zeros(W, _pad8(K), dtype=torch.bfloat16, device=device)
			conv[:, :K].copy_(self.p("visual.trunk.patch_embed.proj.weight").reshape(W, K))
			w16 = {"visual.trunk.patch_embed.proj.weight": conv}
			H, D = cfg.heads, W // cfg.heads
			Dp, Mp = clip_text.padded_head_dim(D), clip_text.padded_mlp_dim(cfg.mlp_dim)
			for n in self.names:
				t = self.p(n)
				# zero padding of heads / of the MLP width (clip_text.padded_head_dim): weights as bf16, the biases that go with padded rows as fp32 copies
				if n.endswith("attn.qkv.weight") or n.endswith("attn.qkv.bias"):
					t = clip_text.pad_rows_per_head(t, 3, H, D, Dp)
				elif n.endswith("attn_pool.kv.weight") or n.endswith("attn_pool.kv.bias"):
					t = clip_text.pad_rows_per_head(t, 2, H, D, Dp)
				elif n.endswith("attn_pool.q.weight") or n.endswith("attn_pool.q.bias"):
					t = clip_text.pad_rows_per_head(t, 1, H, D, Dp)
				elif n.endswith("attn.proj.weight") or n.endswith("attn_pool.proj.weight"):
					t = clip_text.pad_cols_per_head(t, H, D, Dp)
				elif n.endswith("mlp.fc1.weight") or n.endswith("mlp.fc1.bias"):
					t = clip_text.pad_dim(t, 0, Mp)
				elif n.endswith("mlp.fc2.weight"):
					t = clip_text.pad_dim(t, 1, Mp)
				if t.ndim == 2:
					d = torch.empty(t.shape, dtype=torch.bfloat16, device=device)
					ops.cast_bf16(t.contiguous(), d)
					w16[n] = d
				elif t is not self.p(n):
					w16[n] = t.contiguous()
			# q = Linear_q(latent): a constant of the weights -- one [8 x W] x [W x H Dp] GEMM at weight-load time (row 0 is the latent, the rest zero padding)
			a = "visual.trunk.attn_pool."
			lat = torch.zeros(8, W, dtype=torch.bfloat16, device=device)
			lat[0].copy_(self.p(a + "latent").reshape(W))
			q = torch.empty(8, H * Dp, dtype=torch.bfloat16, device=device)
			ops.gemm(lat, w16[a + "q.weight"], 8, H * Dp, W, out=q, bias=w16.get(a + "q.bias", self.p(a + "q.bias")))
			w16["latent_q"] = q[0].clone()
			self._w16, self._w16_key = w16, key
			self._rt_reset()  # (the qkv buffer of the pooling holds the old latent query; captured graphs read the old shadow)
		return self._w16

	def get_image_transform(self):
		pp = self.preprocess or {}
		return make_image_transform(self.cfg.image_size, tuple(pp.get("mean", (0.5, 0.5, 0.5))), tuple(pp.get("std", (0.5, 0.5, 0.5))), pp.get("interpolation", "bicubic"))

	@torch.no_grad()
	def forward(self, images: torch.Tensor, normalize: bool = True) -> torch.Tensor:
		cfg = self.cfg
		if not images.is_cuda or not self.p("visual.trunk.norm.weight").is_cuda:
			raise _lib.NovicHipError("NativeSigLIPViT runs on MI355X only: move the model and the image batch to a 'cuda' device (there is no CPU path)")
		assert images.ndim == 4 and images.shape[1] == 3 and images.shape[2] == images.shape[3] == cfg.image_size and images.dtype == torch.float32
		self._shadow(images.device)  # (first: a weight reload drops the slots, whose graphs read the old bf16 shadow)
		# hipGraph replay per batch shape (tower_runtime.TowerRuntime; the trunk was the one tower still launched eagerly from Python until round 4).  The capture starts
		# BEHIND im2col, which reads the caller's images and runs in front of every replay.
		return self._rt_forward(images, normalize, eager=lambda im: self._forward_lane(im, normalize, False), capture_tail=lambda im: self._forward_lane(im, normalize, True),
		                        before_replay=self._im2col)

	def _im2col(self, images: torch.Tensor) -> torch.Tensor:
		cfg = self.cfg
		Kp = self._shadow(images.device)["visual.trunk.patch_embed.proj.weight"].shape[1]
		patches = self._buf("patches", (images.shape[0] * cfg.tokens, Kp), torch.bfloat16, images.device)
		ops.vit_im2col(images.contiguous(), patches, cfg.patch_size)
		return patches

	def _forward_lane(self, images: torch.Tensor, normalize: bool, skip_im2col: bool) -> torch.Tensor:
		with self._lane_scratch(0, images.device):  # (the K-split scratch of this slot: tower_runtime)
			return self._launches(images, normalize, skip_im2col)

	def _launches(self, images: torch.Tensor, normalize: bool, skip_im2col: bool) -> torch.Tensor:
		cfg = self.cfg
		dev = images.device
		w16 = self._shadow(dev)
		B, W, N, H = images.shape[0], cfg.width, cfg.tokens, cfg.heads
		D = W // H
		Dp, M = clip_text.padded_head_dim(D), clip_text.padded_mlp_dim(cfg.mlp_dim)  # the widths the kernels run (zero-padded weights: _shadow)
		Wp = H * Dp
		scale = None if Dp == D else float(D) ** -0.5
		act = ops.ACT_GELU_TANH if cfg.gelu_tanh else ops.ACT_GELU
		fb = lambda name: w16.get(name, self.p(name))  # an fp32 bias: its padded copy where rows were padded
		T = B * N
		t = "visual.trunk."
		Kp = w16[t + "patch_embed.proj.weight"].shape[1]
		b = lambda name, shape, dtype: self._buf2(name, shape, dtype, dev)
		patches, _ = b("patches", (T, Kp), torch.bfloat16)
		if not skip_im2col:  # (a captured graph starts behind this launch)
			ops.vit_im2col(images.contiguous(), patches, cfg.patch_size)
		pos, fresh = b("pos_tiled", (T, W), torch.float32)
		if fresh:
			pos.view(B, N, W).copy_(self.p(t + "pos_embed").expand(B, N, W))
		x, _ = b("x0", (T, W), torch.float32)
		ops.gemm(patches, w16[t + "patch_embed.proj.weight"], T, W, Kp, kind=ops.EPI_RESID_F32, out=x, resid=pos, bias=self.p(t + "patch_embed.proj.bias"))
		ln, _ = b("ln", (T, W), torch.bfloat16)
		qkv, _ = b("qkv", (T, 3 * Wp), torch.bfloat16)
		att, _ = b("att", (T, Wp), torch.bfloat16)
		hid, _ = b("hid", (T, M), torch.bfloat16)
		x2 = x  # the fp32 residual stream is updated in place (clip_vit.NativeViT.inplace_residual: bit-identical, the second buffer only cost cache)
		for i in range(cfg.layers):
			q = f"{t}blocks.{i}."
			ops.layernorm_fwd(x, self.p(q + "norm1.weight"), ln, T, W, beta=self.p(q + "norm1.bias"), eps=cfg.ln_eps)
			ops.gemm(ln, w16[q + "attn.qkv.weight"], T, 3 * Wp, W, out=qkv, bias=fb(q + "attn.qkv.bias"), split_tail=True)
			ops.vit_attn_fwd(qkv, att, B, N, H, Dp, scale=scale)
			ops.gemm(att, w16[q + "attn.proj.weight"], T, W, Wp, kind=ops.EPI_RESID_F32, out=x2, resid=x, bias=self.p(q + "attn.proj.bias"), split_tail=True)
			ops.layernorm_fwd(x2, self.p(q + "norm2.weight"), ln, T, W, beta=self.p(q + "norm2.bias"), eps=cfg.ln_eps)
			ops.gemm(ln, w16[q + "mlp.fc1.weight"], T, M, W, out=hid, bias=fb(q + "mlp.fc1.bias"), act=act, split_tail=True)
			ops.gemm(hid, w16[q + "mlp.fc2.weight"], T, W, M, kind=ops.EPI_RESID_F32, out=x, resid=x2, bias=self.p(q + "mlp.fc2.bias"), split_tail=True)
		ops.layernorm_fwd(x, self.p(t + "norm.weight"), ln, T, W, beta=self.p(t + "norm.bias"), eps=cfg.ln_eps)
		# attention pooling: K / V of every token into columns W .. 3W of a qkv buffer whose Q columns hold the projected latent
		a = t + "attn_pool."
		pq, fresh = b("pool_qkv", (T, 3 * Wp), torch.bfloat16)
		if fresh:
			pq[:, :Wp].copy_(w16["latent_q"].unsqueeze(0).expand(T, Wp))
		ops.gemm(ln, w16[a + "kv.weight"], T, 2 * Wp, W, out=pq[:, Wp:], ldc=3 * Wp, bias=fb(a + "kv.bias"))
		ops.vit_attn_fwd(pq, att, B, N, H, Dp, scale=scale)
		y, _ = b("pool_y", (B, W), torch.float32)
		ops.gemm(att, w16[a + "proj.weight"], B, W, Wp, lda=N * Wp, kind=ops.EPI_STORE_F32, out=y, bias=self.p(a + "proj.bias"))  # row 0 of every image: lda = N * Wp
		yl, _ = b("pool_ln", (B, W), torch.bfloat16)
		ops.layernorm_fwd(y, self.p(a + "norm.weight"), yl, B, W, beta=self.p(a + "norm.bias"), eps=cfg.ln_eps)
		yh, _ = b("pool_h", (B, M), torch.bfloat16)
		ops.gemm(yl, w16[a + "mlp.fc1.weight"], B, M, W, out=yh, bias=fb(a + "mlp.fc1.bias"), act=act)
		raw = torch.empty((B, W), dtype=torch.float32, device=dev)
		ops.gemm(yh, w16[a + "mlp.fc2.weight"], B, W, M, kind=ops.EPI_RESID_F32, out=raw, resid=y, bias=self.p(a + "mlp.fc2.bias"))
		if not normalize:
			return raw
		out = torch.empty_like(raw)
		ops.rownorm_f32(raw, out)
		return out


def vision_config_from(vc: dict, sd: dict) -> SigLIPVisionConfig:
	"""open_clip `vision_cfg` of a timm-trunk model + its tensors -> dimensions."""
	t = "visual.trunk."
	if vc.get("timm_pool", "map") != "map" or vc.get("timm_proj", "none") not in ("none", "", None):
		raise NotImplementedError(f"timm trunk with timm_pool = {vc.get('timm_pool')!r} / timm_proj = {vc.get('timm_proj')!r}: only the SigLIP form (attention pooling, no projection) is built")
	if t + "cls_token" in sd or t + "attn_pool.latent" not in sd:
		raise NotImplementedError("timm trunk with a class token / without an attention-pool head is not the SigLIP form this tower implements")
	conv = sd[t + "patch_embed.proj.weight"]
	W, p = conv.shape[0], conv.shape[-1]
	layers = len({k.split(".")[3] for k in sd if k.startswith(t + "blocks.")})
	grid = round(sd[t + "pos_embed"].shape[1] ** 0.5)
	name = str(vc.get("timm_model_name", ""))
	heads = vc.get("heads") or next((h for stem, h in TIMM_HEADS.items() if name.startswith(stem)), None) or max(1, W // 64)
	img = vc.get("image_size", grid * p)
	img = int(img[0] if isinstance(img, (list, tuple)) else img)
	if img != grid * p:
		raise ValueError(f"vision_cfg.image_size {img} does not match the positional embedding ({grid} x {grid} patches of {p})")
	return SigLIPVisionConfig(image_size=img, patch_size=p, width=W, layers=layers, heads=int(heads), mlp_dim=sd[t + "blocks.0.mlp.fc1.weight"].shape[0],
	                          gelu_tanh=(vc.get("act_kwargs") or {}).get("approximate") == "tanh")


def build_towers(mc: dict, sd: dict):
	"""open_clip `model_cfg` of a SigLIP model + state dict -> (NativeSigLIPViT, text tower)."""
	vc, tc, F = mc["vision_cfg"], mc.get("text_cfg", {}), int(mc["embed_dim"])
	vcfg = vision_config_from(vc, sd)
	if vcfg.embed_dim != F:
		raise NotImplementedError(f"embed_dim {F} differs from the trunk width {vcfg.width}: a timm projection head is not built")
	vit = NativeSigLIPViT(vcfg)
	vit.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual.trunk.")})
	if tc.get("hf_model_name") or tc.get("embed_cls"):
		raise NotImplementedError("open_clip text_cfg with a Hugging Face text model / CLS embedding is not implemented by the native text tower")
	tw = int(tc.get("width", 768))
	tcfg = clip_text.TextConfig(vocab_size=int(tc.get("vocab_size", 32000)), context_length=int(tc.get("context_length", 64)), width=tw, layers=int(tc.get("layers", 12)),
	                            heads=int(tc.get("heads", 12)), mlp_ratio=float(tc.get("mlp_ratio", 4.0)), embed_dim=F, quick_gelu=False,
	                            ln_eps=float((tc.get("norm_kwargs") or {}).get("eps", 1e-5)), causal=not tc.get("no_causal_mask", False), pool=str(tc.get("pool_type", "argmax")),
	                            proj_bias=bool(tc.get("proj_bias", False)), pad_id=int(tc.get("pad_id", 0)),
	                            gelu_tanh=(tc.get("act_kwargs") or {}).get("approximate") == "tanh")
	txt = clip_text.NativeTextTower(tcfg)
	tsd = {}
	for k, v in sd.items():
		if not k.startswith("text."):
			continue
		k = k[5:]
		if k == "text_projection.weight":
			tsd["text_projection"] = v.T.contiguous()  # nn.Linear [F, W] -> the tower's [W, F]
		elif k == "text_projection.bias":
			tsd["text_projection_bias"] = v
		elif re.match(r"(token_embedding\.|positional_embedding|transformer\.|ln_final\.|text_projection$)", k):
			tsd[k] = v
	txt.load_state_dict(tsd)
	return vit, txt
