"""CLIP text tower on hand-written HIP kernels: the `encode_text` half of the reference's ``inference_tokens`` / ``inference_text``
(embedders.py:423-426, :557-583, :728-753), which the reference delegates to open_clip / clip / transformers.  It is what fills the
embedding cache that decoder training reads (embedding_cache_writers.py:246-356).

``NativeTextTower(token_ids)``: B x S integer ids (S <= context_length; CLIP layout <start> ... <end>, padded) on the device -> B x F fp32
unit rows.  Weights use OpenCLIP's state-dict names (``token_embedding.weight``, ``positional_embedding``, ``transformer.resblocks.N.*``,
``ln_final.*``, ``text_projection``); ``load_hf_state_dict`` maps Hugging Face ``CLIPTextModelWithProjection`` names onto them.  bf16 MFMA GEMMs with
fp32 accumulation, fp32 LayerNorm / softmax / residual stream.  Tokenisation itself (the BPE vocabulary) is host-side data the embedder must be given.

Kernel sequence per batch (all launches, no torch arithmetic): embed(tok + pos) -> L x [LN -> GEMM qkv(+bias) -> causal attention -> GEMM out(+bias,
+residual) -> LN -> GEMM fc1(+bias,+GELU|QuickGELU) -> GEMM fc2(+bias,+residual)] -> gather the END-OF-TEXT rows -> LN -> GEMM projection -> L2 normalise.
"""
from __future__ import annotations

import dataclasses
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .tower_runtime import TowerRuntime


@dataclasses.dataclass(frozen=True)
class TextConfig:
	vocab_size: int = 49408
	context_length: int = 77
	width: int = 512
	layers: int = 12
	heads: int = 8
	mlp_ratio: float = 4.0
	embed_dim: int = 512
	quick_gelu: bool = False
	ln_eps: float = 1e-5
	# open_clip `text_cfg` variants (SigLIP: no_causal_mask, pool_type 'last', proj_bias; novic_amd/siglip.py): attention without the causal mask -- the rows are then
	# padded to the full context with pad_id, as open_clip's tokenizer call does, because the padding takes part --, pooling at the LAST position instead of the
	# END-OF-TEXT token, a bias on the projection
	causal: bool = True
	pool: str = "argmax"
	proj_bias: bool = False
	pad_id: int = 0
	gelu_tanh: bool = False  # text_cfg.act_kwargs.approximate = 'tanh'

	@property
	def mlp_dim(self) -> int:
		return int(self.width * self.mlp_ratio)

	def flops_per_text(self, S: Optional[int] = None) -> float:
		"""Per text of S tokens: L * (S * (8 W^2 + 4 W M) + 4 S^2 W / 2 (causal)) + 2 W F."""
		S = S or self.context_length
		W = self.width
		return self.layers * (S * (8 * W * W + 4 * W * self.mlp_dim) + 2 * S * S * W) + 2 * W * self.embed_dim


# Heads of a width the attention kernels do not have (ViT-SO400M-14-SigLIP: 1152 / 16 = 72) run ZERO-PADDED to the next width they have: the padded rows of the
# in-projection (and its bias) are zeros, so the extra q / k / v columns are zeros, add nothing to q . k and produce zero outputs, which meet zero columns of the
# padded out-projection; only the soft-max scale keeps the real width.  The same trick rounds an MLP width up to the GEMM's K-tile (4304 -> 4352: act(0 + 0) = 0 for
# every activation here): exact, and both MLP GEMMs stay on the 256-wide kernel.
HEAD_DIMS = (32, 64, 80)


def padded_head_dim(D: int) -> int:
	for d in HEAD_DIMS:
		if D <= d:
			return d
	raise NotImplementedError(f"head_dim {D} exceeds the attention kernels' largest width ({HEAD_DIMS[-1]})")


def pad_rows_per_head(t: torch.Tensor, groups: int, H: int, D: int, Dp: int) -> torch.Tensor:
	"""[groups * H * D, ...] -> [groups * H * Dp, ...]: zeros behind each head's D rows."""
	if D == Dp:
		return t
	rest = tuple(t.shape[1:])
	o = t.new_zeros((groups, H, Dp) + rest)
	o[:, :, :D] = t.reshape((groups, H, D) + rest)
	return o.reshape((groups * H * Dp,) + rest)


def pad_cols_per_head(t: torch.Tensor, H: int, D: int, Dp: int) -> torch.Tensor:
	"""[N, H * D] -> [N, H * Dp]."""
	if D == Dp:
		return t
	o = t.new_zeros((t.shape[0], H, Dp))
	o[:, :, :D] = t.reshape(t.shape[0], H, D)
	return o.reshape(t.shape[0], H * Dp)


def pad_dim(t: torch.Tensor, dim: int, n: int) -> torch.Tensor:
	if t.shape[dim] == n:
		return t
	shape = list(t.shape)
	shape[dim] = n
	o = t.new_zeros(shape)
	o.narrow(dim, 0, t.shape[dim]).copy_(t)
	return o


def padded_mlp_dim(M: int) -> int:
	return (M + 63) // 64 * 64


TEXT_B_32 = TextConfig(49408, 77, 512, 12, 8, 4.0, 512, quick_gelu=True)     # openai:ViT-B/32 text side
TEXT_L_14 = TextConfig(49408, 77, 768, 12, 12, 4.0, 768, quick_gelu=False)   # openclip ViT-L-14 text side
TEXT_H_14 = TextConfig(49408, 77, 1024, 24, 16, 4.0, 1024, quick_gelu=False)  # openclip ViT-H-14 text side


class NativeTextTower(TowerRuntime, nn.Module):

	def __init__(self, cfg: TextConfig, seed: Optional[int] = None, eot_token_id: Optional[int] = None):
		"""eot_token_id None: pool at the arg-max token id (CLIP's vocabulary: END-OF-TEXT = 49407 is the largest id); else at the first occurrence of that id."""
		super().__init__()
		self.cfg = cfg
		self.eot_token_id = eot_token_id
		W, L, F, M = cfg.width, cfg.layers, cfg.embed_dim, cfg.mlp_dim
		if W % cfg.heads or (W // cfg.heads) > HEAD_DIMS[-1] or (W // cfg.heads) % 8 or W % 8 or F % 8 or M % 8:
			raise NotImplementedError("NativeTextTower supports head_dim <= 80 (32 / 64 / 80 natively, others zero-padded) and widths that are multiples of 8")
		g = torch.Generator().manual_seed(seed) if seed is not None else None
		n = lambda *shape, std: nn.Parameter(torch.randn(*shape, generator=g) * std)
		sc = W ** -0.5
		self.names: list[str] = []

		def reg(name: str, param: nn.Parameter):
			self.names.append(name)
			self.register_parameter(name.replace(".", "__"), param)
		reg("token_embedding.weight", n(cfg.vocab_size, W, std=0.02))
		reg("positional_embedding", n(cfg.context_length, W, std=0.01))
		reg("ln_final.weight", nn.Parameter(torch.ones(W))); reg("ln_final.bias", nn.Parameter(torch.zeros(W)))
		reg("text_projection", n(W, F, std=sc))
		if cfg.proj_bias:
			reg("text_projection_bias", nn.Parameter(torch.zeros(F)))
		if cfg.pool not in ("argmax", "first", "last"):
			raise NotImplementedError(f"text pooling '{cfg.pool}' is not implemented (argmax / first END token, last position)")
		for i in range(L):
			q = f"transformer.resblocks.{i}."
			for nm in ("ln_1", "ln_2"):
				reg(q + nm + ".weight", nn.Parameter(torch.ones(W))); reg(q + nm + ".bias", nn.Parameter(torch.zeros(W)))
			reg(q + "attn.in_proj_weight", n(3 * W, W, std=sc)); reg(q + "attn.in_proj_bias", nn.Parameter(torch.zeros(3 * W)))
			reg(q + "attn.out_proj.weight", n(W, W, std=sc * (2 * L) ** -0.5)); reg(q + "attn.out_proj.bias", nn.Parameter(torch.zeros(W)))
			reg(q + "mlp.c_fc.weight", n(M, W, std=(2 * W) ** -0.5)); reg(q + "mlp.c_fc.bias", nn.Parameter(torch.zeros(M)))
			reg(q + "mlp.c_proj.weight", n(W, M, std=sc * (2 * L) ** -0.5)); reg(q + "mlp.c_proj.bias", nn.Parameter(torch.zeros(W)))
		for prm in self.parameters():
			prm.requires_grad_(False)
		self._w16: dict[str, torch.Tensor] = {}
		self._w16_key = None

	def p(self, name: str) -> torch.Tensor:
		return getattr(self, name.replace(".", "__"))

	def state_dict(self, *args, **kwargs):
		return {n: self.p(n).detach() for n in self.names}

	def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
		missing = [n for n in self.names if n not in state_dict]
		extra = [k for k in state_dict if k not in self.names]
		if strict and (missing or extra):
			raise RuntimeError(f"NativeTextTower.load_state_dict: missing {missing[:5]}, unexpected {extra[:5]}")
		with torch.no_grad():
			for n in self.names:
				if n in state_dict:
					self.p(n).copy_(state_dict[n])
		self._w16_key = None

	def load_hf_state_dict(self, hf: dict):
		"""Hugging Face CLIPTextModelWithProjection names -> OpenCLIP names."""
		sd = {
			"token_embedding.weight": hf["text_model.embeddings.token_embedding.weight"],
			"positional_embedding": hf["text_model.embeddings.position_embedding.weight"],
			"ln_final.weight": hf["text_model.final_layer_norm.weight"], "ln_final.bias": hf["text_model.final_layer_norm.bias"],
			"text_projection": hf["text_projection.weight"].T,
		}
		for i in range(self.cfg.layers):
			o, h = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
			sd[o + "attn.in_proj_weight"] = torch.cat([hf[h + f"self_attn.{k}_proj.weight"] for k in "qkv"], dim=0)
			sd[o + "attn.in_proj_bias"] = torch.cat([hf[h + f"self_attn.{k}_proj.bias"] for k in "qkv"], dim=0)
			sd[o + "attn.out_proj.weight"], sd[o + "attn.out_proj.bias"] = hf[h + "self_attn.out_proj.weight"], hf[h + "self_attn.out_proj.bias"]
			sd[o + "ln_1.weight"], sd[o + "ln_1.bias"] = hf[h + "layer_norm1.weight"], hf[h + "layer_norm1.bias"]
			sd[o + "ln_2.weight"], sd[o + "ln_2.bias"] = hf[h + "layer_norm2.weight"], hf[h + "layer_norm2.bias"]
			sd[o + "mlp.c_fc.weight"], sd[o + "mlp.c_fc.bias"] = hf[h + "mlp.fc1.weight"], hf[h + "mlp.fc1.bias"]
			sd[o + "mlp.c_proj.weight"], sd[o + "mlp.c_proj.bias"] = hf[h + "mlp.fc2.weight"], hf[h + "mlp.fc2.bias"]
		self.load_state_dict(sd)

	def _shadow(self, device) -> dict[str, torch.Tensor]:
		"""bf16 copies of the GEMM weights, rebuilt when parameters change (the embedding tables stay fp32: they are gathered, not multiplied)."""
		def ver(t):
			try:
				return t._version
			except RuntimeError:
				return 0
		key = (device, tuple(ver(self.p(n)) for n in self.names))
		if self._w16_key != key:
			cfg = self.cfg
			H, D, M = cfg.heads, cfg.width // cfg.heads, cfg.mlp_dim
			Dp, Mp = padded_head_dim(D), padded_mlp_dim(M)
			w16 = {}
			for n in self.names:
				t = self.p(n)
				# zero padding of heads / of the MLP width (see padded_head_dim): weights as bf16, the biases that go with padded rows as fp32 copies
				if n.endswith("attn.in_proj_weight") or n.endswith("attn.in_proj_bias"):
					t = pad_rows_per_head(t, 3, H, D, Dp)
				elif n.endswith("attn.out_proj.weight"):
					t = pad_cols_per_head(t, H, D, Dp)
				elif n.endswith("mlp.c_fc.weight") or n.endswith("mlp.c_fc.bias"):
					t = pad_dim(t, 0, Mp)
				elif n.endswith("mlp.c_proj.weight"):
					t = pad_dim(t, 1, Mp)
				if t.ndim == 2 and n not in ("positional_embedding", "token_embedding.weight"):
					d = torch.empty(t.shape, dtype=torch.bfloat16, device=device)
					ops.cast_bf16(t.contiguous(), d)
					w16[n] = d
				elif t is not self.p(n):
					w16[n] = t.contiguous()
			self._w16, self._w16_key = w16, key
			self._rt_reset()  # captured graphs read the old shadow's buffers
		return self._w16

	# Lanes: as NativeViT.forward -- sub-batches on streams of their own fill the partly empty last rounds of each other's persistent GEMM grids (off by default, see there).
	half_stream = False  # the residual stream as IEEE half instead of fp32: local_clip.OpenAIEmbedder switches it on (the reference runs that family as clip's fp16 model)
	lanes = 1
	lane_min_rows = 32768  # token rows a lane must keep (see NativeViT: smaller sub-batches lose more to tile selection than the overlap gains)

	@torch.no_grad()
	def forward(self, token_ids: torch.Tensor, normalize: bool = True) -> torch.Tensor:
		cfg = self.cfg
		if not token_ids.is_cuda or not self.p("text_projection").is_cuda:
			raise _lib.NovicHipError("NativeTextTower runs on MI355X only: move the model and the token batch to a 'cuda' device (there is no CPU path)")
		assert token_ids.ndim == 2 and token_ids.dtype in (torch.int32, torch.int64) and 1 <= token_ids.shape[1] <= cfg.context_length
		n_lanes = max(1, min(int(self.lanes), token_ids.shape[0] * token_ids.shape[1] // max(1, int(self.lane_min_rows))))
		if n_lanes <= 1:
			return self._forward_graphed(token_ids, normalize)
		dev = token_ids.device
		self._shadow(dev)
		main = torch.cuda.current_stream(dev)
		pool = ops.lane_streams(dev, n_lanes)
		B = token_ids.shape[0]
		edges = [B * i // n_lanes for i in range(n_lanes + 1)]
		out = torch.empty((B, cfg.embed_dim), dtype=torch.float32, device=dev)
		with self._rt_use(self._rt_slot(("lanes", n_lanes, tuple(token_ids.shape), bool(normalize), dev), dev)):
			for i in range(n_lanes):
				st = pool[i]
				st.wait_stream(main)
				with torch.cuda.stream(st):
					out[edges[i]:edges[i + 1]].copy_(self._forward_lane(token_ids[edges[i]:edges[i + 1]], normalize, i))
		for st in pool[:n_lanes]:
			main.wait_stream(st)
		return out

	def _forward_graphed(self, token_ids: torch.Tensor, normalize: bool) -> torch.Tensor:
		"""hipGraph replay per batch shape (tower_runtime.TowerRuntime); the ids are copied into the slot's static input in front of every replay."""
		self._shadow(token_ids.device)  # (first: a weight reload drops the slots, whose graphs read the old bf16 shadow)
		return self._rt_forward(token_ids, normalize, eager=lambda ids: self._forward_lane(ids, normalize, 0), static_input=True)

	def _forward_lane(self, token_ids: torch.Tensor, normalize: bool, lane: int) -> torch.Tensor:
		with self._lane_scratch(lane, token_ids.device):  # (the K-split scratch of this slot and lane: tower_runtime)
			return self._launches(token_ids, normalize, lane)

	def _launches(self, token_ids: torch.Tensor, normalize: bool, lane: int) -> torch.Tensor:
		cfg = self.cfg
		dev = token_ids.device
		w16 = self._shadow(dev)
		if not cfg.causal and token_ids.shape[1] < cfg.context_length:  # without a causal mask every position sees the padding: run the full context, as open_clip does
			full = token_ids.new_full((token_ids.shape[0], cfg.context_length), cfg.pad_id)
			full[:, :token_ids.shape[1]] = token_ids
			token_ids = full
		B, S = token_ids.shape
		W, H, F = cfg.width, cfg.heads, cfg.embed_dim
		D = W // H
		Dp, M = padded_head_dim(D), padded_mlp_dim(cfg.mlp_dim)  # the widths the kernels run (zero-padded weights: _shadow)
		Wp = H * Dp
		scale = None if Dp == D else float(D) ** -0.5
		T = B * S
		ids = token_ids.contiguous()
		b = lambda name, shape, dtype: self._buf(f"L{lane}:{name}", shape, dtype, dev)
		fb = lambda name: w16.get(name, self.p(name))  # an fp32 bias: its padded copy where rows were padded
		half = bool(self.half_stream)  # the residual stream as IEEE half: clip's fp16 text tower ('openai:' embedders only; clip_vit.NativeViT.half_stream has the why)
		sdt, resid_kind = (torch.float16, ops.EPI_RESID_F16) if half else (torch.float32, ops.EPI_RESID_F32)
		x = b("x0h" if half else "x0", (T, W), sdt)
		x2 = x  # the residual stream is updated in place (clip_vit.NativeViT.inplace_residual: bit-identical, the second buffer only cost cache)
		ops.text_embed(ids, self.p("token_embedding.weight"), self.p("positional_embedding"), x, B, S, W)
		ln, qkv, att, hid = b("ln", (T, W), torch.bfloat16), b("qkv", (T, 3 * Wp), torch.bfloat16), b("att", (T, Wp), torch.bfloat16), b("hid", (T, M), torch.bfloat16)
		act = ops.ACT_QUICKGELU if cfg.quick_gelu else ops.ACT_GELU_TANH if cfg.gelu_tanh else ops.ACT_GELU
		for i in range(cfg.layers):
			q = f"transformer.resblocks.{i}."
			ops.layernorm_fwd(x, self.p(q + "ln_1.weight"), ln, T, W, beta=self.p(q + "ln_1.bias"), eps=cfg.ln_eps)
			ops.gemm(ln, w16[q + "attn.in_proj_weight"], T, 3 * Wp, W, out=qkv, bias=fb(q + "attn.in_proj_bias"), split_tail=True)
			ops.clip_attn_fwd(qkv, att, B, S, H, Dp, causal=cfg.causal, scale=scale)
			ops.gemm(att, w16[q + "attn.out_proj.weight"], T, W, Wp, kind=resid_kind, out=x2, resid=x, bias=self.p(q + "attn.out_proj.bias"), split_tail=True)
			ops.layernorm_fwd(x2, self.p(q + "ln_2.weight"), ln, T, W, beta=self.p(q + "ln_2.bias"), eps=cfg.ln_eps)
			ops.gemm(ln, w16[q + "mlp.c_fc.weight"], T, M, W, out=hid, bias=fb(q + "mlp.c_fc.bias"), act=act, split_tail=True)
			ops.gemm(hid, w16[q + "mlp.c_proj.weight"], T, W, M, kind=resid_kind, out=x, resid=x2, bias=self.p(q + "mlp.c_proj.bias"), split_tail=True)
		pl = b("pooled_ln", (B, W), torch.bfloat16)
		if cfg.pool == "last":  # the final norm of position S - 1 of every row
			ops.layernorm_fwd(x, self.p("ln_final.weight"), pl, B, W, beta=self.p("ln_final.bias"), seq_in=S, seq_out=1, seq_off=S - 1, eps=cfg.ln_eps)
		else:
			pooled = b("pooled", (B, W), torch.float32)
			ops.text_pool(ids, x, pooled, B, S, W, -1 if self.eot_token_id is None else int(self.eot_token_id))
			ops.layernorm_fwd(pooled, self.p("ln_final.weight"), pl, B, W, beta=self.p("ln_final.bias"), eps=cfg.ln_eps)
		raw = torch.empty((B, F), dtype=torch.float32, device=dev)
		ops.gemm(pl, w16["text_projection"], B, F, W, b_kstrided=True, kind=ops.EPI_STORE_F32, out=raw, bias=self.p("text_projection_bias") if cfg.proj_bias else None)
		if not normalize:
			return raw
		out = torch.empty_like(raw)
		ops.rownorm_f32(raw, out)
		return out
