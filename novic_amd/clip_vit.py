"""CLIP ViT image tower on hand-written HIP kernels: the `encode_image` half of the reference's ``inference_image``
(embedders.py:589-594, :759-764, :902-907), which the reference delegates to open_clip / clip / transformers.

``NativeViT(images)``: B x 3 x R x R fp32 (already preprocessed) on the device -> B x F fp32 unit rows.  Weights use OpenCLIP's
``visual.*`` state-dict names; ``load_hf_state_dict`` maps Hugging Face ``CLIPVisionModelWithProjection`` names onto them, so a
locally saved OpenCLIP / OpenAI / HF checkpoint file (no network) can be loaded.  bf16 MFMA GEMMs with fp32 accumulation, fp32
LayerNorm / softmax / residual stream -- what the reference runs under its embedder autocast.

Kernel sequence per batch (all launches, no torch arithmetic): im2col -> GEMM(conv1) -> embed(+cls,+pos, ln_pre) -> L x [LN -> GEMM
qkv(+bias) -> attention -> GEMM out(+bias,+residual) -> LN -> GEMM fc1(+bias,+GELU|QuickGELU) -> GEMM fc2(+bias,+residual)] ->
LN(cls rows) -> GEMM proj -> L2 normalise.
"""
from __future__ import annotations

import dataclasses
import math
from typing import Optional

import torch
import torch.nn as nn

from . import _lib, ops
from .tower_runtime import TowerRuntime

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclasses.dataclass(frozen=True)
class ViTConfig:
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
	def tokens(self) -> int:
		return (self.image_size // self.patch_size) ** 2 + 1

	@property
	def mlp_dim(self) -> int:
		return int(self.width * self.mlp_ratio)

	def flops_per_image(self) -> float:
		"""L*(N*24W^2 + 4N^2 W) + 2*(3p^2)*W*(N-1) + 2*W*F (mlp_ratio 4)."""
		N, W = self.tokens, self.width
		return self.layers * (N * (8 * W * W + 4 * W * self.mlp_dim) + 4 * N * N * W) + 2 * 3 * self.patch_size ** 2 * W * (N - 1) + 2 * W * self.embed_dim


VIT_B_32 = ViTConfig(224, 32, 768, 12, 12, 4.0, 512, quick_gelu=True)    # openai:ViT-B/32 (F = 512)
VIT_L_14 = ViTConfig(224, 14, 1024, 24, 16, 4.0, 768, quick_gelu=False)  # openclip ViT-L-14 (DataComp / DFN2B)
VIT_H_14 = ViTConfig(224, 14, 1280, 32, 16, 4.0, 1024, quick_gelu=False)  # openclip ViT-H-14 (DFN5B / laion2B)


def make_image_transform(R: int, mean=CLIP_MEAN, std=CLIP_STD, interpolation: str = "bicubic", uint8: bool = False):
	"""open_clip 2.23 `image_transform(is_train=False)` = torchvision Resize(R, interpolation) -> CenterCrop(R) -> RGB -> ToTensor -> Normalize(mean, std), restated on PIL
	(embedders.py:755-757 returns exactly that callable).  Resize(int) scales the SHORTER side to R and the longer one to int(R * long / short) -- truncated, not rounded
	(torchvision `_compute_resized_output_size`); CenterCrop takes offsets round((size - R) / 2)."""
	mean_t, std_t = torch.tensor(mean, dtype=torch.float32).view(3, 1, 1), torch.tensor(std, dtype=torch.float32).view(3, 1, 1)

	def transform(img):
		import numpy as np
		from PIL import Image
		resample = {"bicubic": Image.BICUBIC, "bilinear": Image.BILINEAR, "nearest": Image.NEAREST}[interpolation]
		w, h = img.size
		if min(w, h) != R:
			if w <= h:
				nw, nh = R, int(R * h / w)
			else:
				nw, nh = int(R * w / h), R
			img = img.resize((nw, nh), resample)
		w, h = img.size
		l, t = int(round((w - R) / 2.0)), int(round((h - R) / 2.0))
		img = img.crop((l, t, l + R, t + R)).convert("RGB")
		if uint8:  # the pixels before ToTensor / Normalize, 3 x R x R uint8: the tower's im2col applies both steps on the device (novic_vit_im2col_u8), a quarter of the bytes per image
			return torch.from_numpy(np.array(img, dtype=np.uint8)).permute(2, 0, 1).contiguous()
		arr = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
		return (arr - mean_t) / std_t
	return transform


def _pad8(n: int) -> int:
	return (n + 7) // 8 * 8


class NativeViT(TowerRuntime, nn.Module):
	# (Round 4 built the LayerNorm FOLDED into the GEMMs around each residual add -- proj / fc2 also wrote a bf16 copy of the residual stream and per-row partial sums, QKV /
	# fc1 multiplied that copy with gamma-scaled weights and normalised in their epilogues: 23 of ViT-B/32's 24 LayerNorm launches gone, exact up to where bf16 rounding
	# happens -- and measured it slower, 69.1 k -> 68.8 k images/s at ViT-B/32 batch 256, 5 518 -> 5 173 at ViT-L/14: the bytes the LayerNorm launch moved at the full-chip
	# rate moved inside single-round GEMM epilogues that were HBM-bound already.  Removed in round 5 with its ABI fields: DESIGN.md section 4, "Round 4" / "Round 5".)
	# The fp32 residual stream is updated IN PLACE by the out-projection / fc2 epilogues (out = resid: every element is read and written by the same lane, once): the lines a
	# tile writes are the lines it has just read, instead of a second 39 MB buffer (ViT-B/32, batch 256) pushing the first out of the Infinity Cache between two uses.
	inplace_residual = True
	share_buffers = False  # (ln / att and qkv / hid in two buffers instead of four: measured the same, tools/inplace_ab.py VIT_B_32 256 share_buffers)
	# Round 6: the residual stream as IEEE HALF.  The reference runs the OpenAI-CLIP family in fp16 END TO END (`OpenAIEmbedder`: manual_amp_dtype = torch.float16,
	# embedders.py:488-489 -- clip's own half-precision model: fp16 weights and activations, LayerNorm computed in fp32 and cast back, residual adds in fp16), so for
	# those towers the fp32 stream kept so far is stricter than the reference and a third of the tower's HBM bytes: per layer the two residual epilogues read and write it
	# and the two LayerNorms read it (6 x rows x W x 4 bytes).  With `half_stream` the stream is fp16 -- ops.EPI_RESID_F16: out = f16(resid + f16(acc + bias)), LayerNorm and
	# vit_embed read / write half, statistics in fp32 -- while the GEMM operands stay bf16 (the reference's are fp16: three mantissa bits more; inside the tower tolerance,
	# tests/test_gpu_vit.py).  `local_clip` switches it on for 'openai:' specs only: open_clip / SigLIP / transformers towers run under autocast with fp32 parameters and an
	# fp32 stream in the reference too.  (bf16 would lose three bits per residual add over twelve layers: outside the tolerance; it has to be the reference's fp16.)
	half_stream = False

	def __init__(self, cfg: ViTConfig, seed: Optional[int] = None):
		super().__init__()
		self.cfg = cfg
		W, L, F, M, p = cfg.width, cfg.layers, cfg.embed_dim, cfg.mlp_dim, cfg.patch_size
		if W % cfg.heads or (W // cfg.heads) not in (32, 64, 80) or W % 8 or F % 8:
			raise NotImplementedError("NativeViT supports head_dim 32/64/80 and widths that are multiples of 8")
		g = torch.Generator().manual_seed(seed) if seed is not None else None
		n = lambda *shape, std: nn.Parameter(torch.randn(*shape, generator=g) * std)
		sc = W ** -0.5
		self.names: list[str] = []

		def reg(name: str, param: nn.Parameter):
			self.names.append(name)
			self.register_parameter(name.replace(".", "__"), param)
		reg("visual.conv1.weight", n(W, 3, p, p, std=0.02))
		reg("visual.class_embedding", n(W, std=sc))
		reg("visual.positional_embedding", n(cfg.tokens, W, std=sc))
		for nm in ("ln_pre", "ln_post"):
			reg(f"visual.{nm}.weight", nn.Parameter(torch.ones(W)))
			reg(f"visual.{nm}.bias", nn.Parameter(torch.zeros(W)))
		reg("visual.proj", n(W, F, std=sc))
		for i in range(L):
			q = f"visual.transformer.resblocks.{i}."
			for nm in ("ln_1", "ln_2"):
				reg(q + nm + ".weight", nn.Parameter(torch.ones(W)))
				reg(q + nm + ".bias", nn.Parameter(torch.zeros(W)))
			reg(q + "attn.in_proj_weight", n(3 * W, W, std=sc)); reg(q + "attn.in_proj_bias", nn.Parameter(torch.zeros(3 * W)))
			reg(q + "attn.out_proj.weight", n(W, W, std=sc * (2 * L) ** -0.5)); reg(q + "attn.out_proj.bias", nn.Parameter(torch.zeros(W)))
			reg(q + "mlp.c_fc.weight", n(M, W, std=(2 * W) ** -0.5)); reg(q + "mlp.c_fc.bias", nn.Parameter(torch.zeros(M)))
			reg(q + "mlp.c_proj.weight", n(W, M, std=sc * (2 * L) ** -0.5)); reg(q + "mlp.c_proj.bias", nn.Parameter(torch.zeros(W)))
		for prm in self.parameters():
			prm.requires_grad_(False)
		self._w16: dict[str, torch.Tensor] = {}
		self._w16_key = None

	# ---- weights ----
	def p(self, name: str) -> torch.Tensor:
		return getattr(self, name.replace(".", "__"))

	def state_dict(self, *args, **kwargs):
		return {n: self.p(n).detach() for n in self.names}

	def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
		missing = [n for n in self.names if n not in state_dict]
		extra = [k for k in state_dict if k not in self.names]
		if strict and (missing or extra):
			raise RuntimeError(f"NativeViT.load_state_dict: missing {missing[:5]}, unexpected {extra[:5]}")
		with torch.no_grad():
			for n in self.names:
				if n in state_dict:
					self.p(n).copy_(state_dict[n])
		self._w16_key = None

	def load_hf_state_dict(self, hf: dict):
		"""Hugging Face CLIPVisionModelWithProjection names -> OpenCLIP `visual.*` names."""
		W = self.cfg.width
		sd = {
			"visual.class_embedding": hf["vision_model.embeddings.class_embedding"],
			"visual.conv1.weight": hf["vision_model.embeddings.patch_embedding.weight"],
			"visual.positional_embedding": hf["vision_model.embeddings.position_embedding.weight"],
			"visual.ln_pre.weight": hf["vision_model.pre_layrnorm.weight"], "visual.ln_pre.bias": hf["vision_model.pre_layrnorm.bias"],
			"visual.ln_post.weight": hf["vision_model.post_layernorm.weight"], "visual.ln_post.bias": hf["vision_model.post_layernorm.bias"],
			"visual.proj": hf["visual_projection.weight"].T,
		}
		for i in range(self.cfg.layers):
			o, h = f"visual.transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
			sd[o + "attn.in_proj_weight"] = torch.cat([hf[h + f"self_attn.{k}_proj.weight"] for k in "qkv"], dim=0)
			sd[o + "attn.in_proj_bias"] = torch.cat([hf[h + f"self_attn.{k}_proj.bias"] for k in "qkv"], dim=0)
			sd[o + "attn.out_proj.weight"], sd[o + "attn.out_proj.bias"] = hf[h + "self_attn.out_proj.weight"], hf[h + "self_attn.out_proj.bias"]
			sd[o + "ln_1.weight"], sd[o + "ln_1.bias"] = hf[h + "layer_norm1.weight"], hf[h + "layer_norm1.bias"]
			sd[o + "ln_2.weight"], sd[o + "ln_2.bias"] = hf[h + "layer_norm2.weight"], hf[h + "layer_norm2.bias"]
			sd[o + "mlp.c_fc.weight"], sd[o + "mlp.c_fc.bias"] = hf[h + "mlp.fc1.weight"], hf[h + "mlp.fc1.bias"]
			sd[o + "mlp.c_proj.weight"], sd[o + "mlp.c_proj.bias"] = hf[h + "mlp.fc2.weight"], hf[h + "mlp.fc2.bias"]
		self.load_state_dict(sd)

	def _shadow(self, device) -> dict[str, torch.Tensor]:
		"""bf16 copies of the GEMM weights (conv1 flattened and K-padded to a multiple of 8), rebuilt when parameters change."""
		def ver(t):
			try:
				return t._version
			except RuntimeError:  # inference tensors have no version counter
				return 0
		key = (device, tuple(ver(self.p(n)) for n in self.names))
		if self._w16_key != key:
			cfg = self.cfg
			K = 3 * cfg.patch_size ** 2
			conv = torch.zeros(cfg.width, _pad8(K), dtype=torch.bfloat16, device=device)
			conv[:, :K].copy_(self.p("visual.conv1.weight").reshape(cfg.width, K))  # one-off at weight load (K padded to a 16-byte multiple)
			w16 = {"visual.conv1.weight": conv}
			for n in self.names:
				t = self.p(n)
				if t.ndim == 2 and n != "visual.positional_embedding":
					d = torch.empty(t.shape, dtype=torch.bfloat16, device=device)
					ops.cast_bf16(t.contiguous(), d)
					w16[n] = d
			self._w16, self._w16_key = w16, key
			self._rt_reset()  # captured graphs read the old shadow's buffers
		return self._w16

	def get_image_transform(self, uint8: bool = False):
		"""PIL image -> 3 x R x R fp32 tensor: the OpenAI / OpenCLIP inference preprocess (host side), with the mean / std / interpolation of `self.preprocess`
		(an open_clip `preprocess_cfg`; CLIP's constants and bicubic when absent).  uint8 = True: the same up to the crop, the pixels as 3 x R x R uint8 -- forward()
		takes such batches and normalises them on the device with the same fp32 arithmetic."""
		pp = getattr(self, "preprocess", None) or {}
		return make_image_transform(self.cfg.image_size, tuple(pp.get("mean", CLIP_MEAN)), tuple(pp.get("std", CLIP_STD)), pp.get("interpolation", "bicubic"), uint8=uint8)

	# ---- forward ----
	# Lanes: a batch of >= 2 * lane_min_rows token rows is cut into `lanes` sub-batches that run the whole tower on streams of their own (own workspace).  Every GEMM of a tower
	# is a persistent grid of <= 256 workgroups whose last round is partly empty (ViT-B/32 at batch 256: 1.76 / 0.78 / 2.34 / 0.78 rounds of tiles per layer), and the
	# launches of one image batch depend on each other; two independent sub-batches fill each other's idle CUs as workgroups retire.  An image's embedding does not
	# depend on the batch it is in (rows of a GEMM are independent) except through the K-split of tail tiles, whose fp32 summation order follows the tile count.
	# Measured (MI355X, batch 256): ViT-L/14 5.67 k -> 5.87 k images/s with two lanes; ViT-B/32 66.8 k -> 46.8 k and the text tower 79.6 k -> 62.1 k -- half a batch of
	# 50- / 77-token rows falls below the tile counts the 256-wide kernels are chosen for -- so a lane must keep >= lane_min_rows token rows.
	# OFF by default (lanes = 1): the + 3 % at ViT-L/14 hold only while the two lanes happen to share a hardware queue; on queues of their own (GPU_MAX_HW_QUEUES = 8, which the
	# decode lanes need) the two persistent grids take each other's CUs: 5 645 -> 4 951 images/s inside bench.py.  Set `tower.lanes = 2` to try it on another workload.
	lanes = 1
	lane_min_rows = 32768

	@torch.no_grad()
	def forward(self, images: torch.Tensor, normalize: bool = True) -> torch.Tensor:
		cfg = self.cfg
		if isinstance(images, (list, tuple)):  # several batches as one forward (pipeline_image_batches(coalesce = n) hands the tower a list)
			return self.forward_many(images, normalize)
		if not images.is_cuda or not self.p("visual.proj").is_cuda:
			raise _lib.NovicHipError("NativeViT runs on MI355X only: move the model and the image batch to a 'cuda' device (there is no CPU path)")
		assert images.ndim == 4 and images.shape[1] == 3 and images.shape[2] == images.shape[3] == cfg.image_size and images.dtype in (torch.float32, torch.uint8)
		n_lanes = max(1, min(int(self.lanes), images.shape[0] * cfg.tokens // max(1, int(self.lane_min_rows))))
		if n_lanes <= 1 or images.dtype == torch.uint8:
			return self._forward_graphed(images, normalize)
		self._shadow(images.device)  # (the bf16 weight shadow is built once, on the caller's stream, before the lanes read it)
		B = images.shape[0]
		edges = [B * i // n_lanes for i in range(n_lanes + 1)]

		def im2col_all(im):
			for i in range(n_lanes):
				self._im2col(im[edges[i]:edges[i + 1]], i)
		return self._rt_forward(images, normalize, eager=lambda im: self._forward_lanes(im, normalize, edges, False), capture_tail=lambda im: self._forward_lanes(im, normalize, edges, True),
		                        before_replay=im2col_all, variant=("lanes", n_lanes, self.lane_cus))

	# Workgroups each lane's persistent GEMM grids may take (None: 256 / lanes): two full-chip grids on streams of their own take each other's CUs one tile at a time
	# (round 3: 5 645 -> 4 951 images/s); with a budget per lane (ops.cu_budget, a per-call argument since ABI 8) the lanes own disjoint CUs, and one lane's HBM-bound
	# phases -- the fp32-residual epilogues, LayerNorm, attention -- run under the other's K loops.
	lane_cus = None

	def _forward_lanes(self, images: torch.Tensor, normalize: bool, edges, skip_im2col: bool) -> torch.Tensor:
		"""Fork onto one stream per lane, join on the caller's (inside a capture: branches of the graph)."""
		dev = images.device
		n_lanes = len(edges) - 1
		main = torch.cuda.current_stream(dev)
		pool = ops.lane_streams(dev, n_lanes)
		out = torch.empty((images.shape[0], self.cfg.embed_dim), dtype=torch.float32, device=dev)
		cus = int(self.lane_cus) if self.lane_cus else max(8, 256 // n_lanes // 8 * 8)
		for i in range(n_lanes):
			st = pool[i]
			st.wait_stream(main)
			with torch.cuda.stream(st), ops.cu_budget(cus):
				out[edges[i]:edges[i + 1]].copy_(self._forward_lane(images[edges[i]:edges[i + 1]], normalize, i, skip_im2col))
		for st in pool[:n_lanes]:
			main.wait_stream(st)
		return out

	@torch.no_grad()
	def forward_many(self, batches, normalize: bool = True) -> torch.Tensor:
		"""ONE forward over several image batches -> [sum of their sizes][F]: each batch's patches are written into its row range of one patch matrix (im2col reads the
		caller's tensors where they lie: no concatenation of 154 MB batches) and the tower runs once over all rows.  What `Embedder.inference_image_batches(coalesce = n)`
		calls: at ViT-B/32 a caller batch of 256 is 150 tiles of the out-projection / fc2 GEMMs on 256 CUs, four of them are 600 (infer_vit_b32_mfma_frac 0.27 -> 0.33).
		The rows of a GEMM are independent and a row's K order does not depend on the tile it is in, so an image's embedding is the one forward(batch) gives -- bit for
		bit, except through K-split tail tiles, whose fp32 summation order follows the launch's tile count (tests/test_gpu_fullsize_properties.py says which)."""
		batches = list(batches)
		if len(batches) == 1:
			return self.forward(batches[0], normalize)
		cfg = self.cfg
		for im in batches:
			if not im.is_cuda:
				raise _lib.NovicHipError("NativeViT runs on MI355X only: move the image batches to a 'cuda' device (there is no CPU path)")
			assert im.ndim == 4 and im.shape[1] == 3 and im.shape[2] == im.shape[3] == cfg.image_size and im.dtype == batches[0].dtype and im.dtype in (torch.float32, torch.uint8)
		self._shadow(batches[0].device)
		return self._rt_forward(batches, normalize, eager=lambda ims: self._forward_lane(ims, normalize, 0), capture_tail=lambda ims: self._forward_lane(ims, normalize, 0, skip_im2col=True),
		                        before_replay=lambda ims: self._im2col(ims, 0))

	def ksplit_tail_planned(self, n_images: int, cus: Optional[int] = None) -> bool:
		"""Would a forward over n_images run any of its layer GEMMs with a K-split tail (on `cus` workgroups: the budget the pipelined tower is launched with)?  Those tails
		are planned per launch -- their fp32 summation order follows the launch's tile count -- so only launches WITHOUT one give every image the embedding of a single-batch
		call bit for bit; `Embedder.inference_image_batches` coalesces caller batches only then.  Host arithmetic (novic_gemm256_plan: no launch, no GPU)."""
		cfg = self.cfg
		T, W, M = int(n_images) * cfg.tokens, cfg.width, cfg.mlp_dim
		resid = ops.EPI_RESID_F16 if self.half_stream else ops.EPI_RESID_F32
		with ops.cu_budget(cus):
			return any(ops.gemm256_plan(T, N, K, kind=kind, bias=True, split_tail=True)["tail_parts"] != 0
			           for N, K, kind in ((3 * W, W, ops.EPI_STORE_BF16), (W, W, resid), (M, W, ops.EPI_STORE_BF16), (W, M, resid)))

	def _pixel_norm(self):
		pp = getattr(self, "preprocess", None) or {}
		return tuple(pp.get("mean", CLIP_MEAN)), tuple(pp.get("std", CLIP_STD))

	def _forward_graphed(self, images: torch.Tensor, normalize: bool) -> torch.Tensor:
		"""hipGraph replay per batch shape (tower_runtime.TowerRuntime).  The capture starts BEHIND im2col: that launch reads the caller's images and writes the slot's patch
		buffer, so it runs eagerly in front of every replay and the graph needs no static copy of the images (154 MB and 54 us per ViT-B/32 batch of 256 that a copy would cost)."""
		self._shadow(images.device)  # (first: a weight reload drops the slots, whose graphs read the old bf16 shadow)
		return self._rt_forward(images, normalize, eager=lambda im: self._forward_lane(im, normalize, 0), capture_tail=lambda im: self._forward_lane(im, normalize, 0, skip_im2col=True),
		                        before_replay=lambda im: self._im2col(im, 0))

	def _im2col(self, images, lane: int) -> torch.Tensor:
		"""images: a batch, or a list of batches laid out one after the other in the patch matrix (forward_many).  uint8 batches are normalised on the way."""
		cfg = self.cfg
		batches = list(images) if isinstance(images, (list, tuple)) else [images]
		dev = batches[0].device
		Kp = self._shadow(dev)["visual.conv1.weight"].shape[1]
		g2 = cfg.tokens - 1
		patches = self._buf(f"L{lane}:patches", (sum(im.shape[0] for im in batches) * g2, Kp), torch.bfloat16, dev)
		norm = self._pixel_norm() if batches[0].dtype == torch.uint8 else None
		row = 0
		for im in batches:
			ops.vit_im2col(im.contiguous(), patches[row:row + im.shape[0] * g2], cfg.patch_size, norm=norm)
			row += im.shape[0] * g2
		return patches

	def _forward_lane(self, images, normalize: bool, lane: int, skip_im2col: bool = False) -> torch.Tensor:
		with self._lane_scratch(lane, images[0].device if isinstance(images, (list, tuple)) else images.device):  # (the K-split scratch of this slot and lane: tower_runtime)
			return self._launches(images, normalize, lane, skip_im2col)

	def _launches(self, images, normalize: bool, lane: int, skip_im2col: bool) -> torch.Tensor:
		cfg = self.cfg
		many = isinstance(images, (list, tuple))
		dev = images[0].device if many else images.device
		w16 = self._shadow(dev)
		B, W, N, H, M, F = (sum(im.shape[0] for im in images) if many else images.shape[0]), cfg.width, cfg.tokens, cfg.heads, cfg.mlp_dim, cfg.embed_dim
		D = W // H
		T = B * N
		Kp = w16["visual.conv1.weight"].shape[1]
		b = lambda name, shape, dtype: self._buf(f"L{lane}:{name}", shape, dtype, dev)
		patches = b("patches", (B * (N - 1), Kp), torch.bfloat16)
		if not skip_im2col:  # (a captured graph starts behind this launch: _forward_graphed)
			self._im2col(images, lane)
		pe = b("pe", (B * (N - 1), W), torch.bfloat16)
		ops.gemm(patches, w16["visual.conv1.weight"], B * (N - 1), W, Kp, out=pe)
		half = bool(self.half_stream)
		sdt, resid_kind = (torch.float16, ops.EPI_RESID_F16) if half else (torch.float32, ops.EPI_RESID_F32)
		x = b("x0h" if half else "x0", (T, W), sdt)
		ops.vit_embed(pe, self.p("visual.class_embedding"), self.p("visual.positional_embedding"), self.p("visual.ln_pre.weight"), self.p("visual.ln_pre.bias"), x, B, N, W, cfg.ln_eps)
		if self.share_buffers:
			# two buffers for the four bf16 activations: ln / att and qkv / hid are never live together (ln dies in the QKV GEMM, att is born in the attention kernel ...)
			ln = att = b("ln", (T, W), torch.bfloat16)
			wide = b("wide", (T * max(3 * W, M),), torch.bfloat16)
			qkv, hid = wide[: T * 3 * W].view(T, 3 * W), wide[: T * M].view(T, M)
		else:
			ln = b("ln", (T, W), torch.bfloat16)
			qkv = b("qkv", (T, 3 * W), torch.bfloat16)
			att = b("att", (T, W), torch.bfloat16)
			hid = b("hid", (T, M), torch.bfloat16)
		x2 = x if self.inplace_residual else b("x1h" if half else "x1", (T, W), sdt)
		act = ops.ACT_QUICKGELU if cfg.quick_gelu else ops.ACT_GELU
		for i in range(cfg.layers):
			q = f"visual.transformer.resblocks.{i}."
			ops.layernorm_fwd(x, self.p(q + "ln_1.weight"), ln, T, W, beta=self.p(q + "ln_1.bias"), eps=cfg.ln_eps)
			ops.gemm(ln, w16[q + "attn.in_proj_weight"], T, 3 * W, W, out=qkv, bias=self.p(q + "attn.in_proj_bias"), split_tail=True)
			ops.vit_attn_fwd(qkv, att, B, N, H, D)
			ops.gemm(att, w16[q + "attn.out_proj.weight"], T, W, W, kind=resid_kind, out=x2, resid=x, bias=self.p(q + "attn.out_proj.bias"), split_tail=True)
			ops.layernorm_fwd(x2, self.p(q + "ln_2.weight"), ln, T, W, beta=self.p(q + "ln_2.bias"), eps=cfg.ln_eps)
			ops.gemm(ln, w16[q + "mlp.c_fc.weight"], T, M, W, out=hid, bias=self.p(q + "mlp.c_fc.bias"), act=act, split_tail=True)
			ops.gemm(hid, w16[q + "mlp.c_proj.weight"], T, W, M, kind=resid_kind, out=x, resid=x2, bias=self.p(q + "mlp.c_proj.bias"), split_tail=True)
		cls = b("cls", (B, W), torch.bfloat16)
		ops.layernorm_fwd(x, self.p("visual.ln_post.weight"), cls, B, W, beta=self.p("visual.ln_post.bias"), seq_in=N, seq_out=1, seq_off=0, eps=cfg.ln_eps)
		raw = torch.empty((B, F), dtype=torch.float32, device=dev)
		ops.gemm(cls, w16["visual.proj"], B, F, W, b_kstrided=True, kind=ops.EPI_STORE_F32, out=raw)
		if not normalize:
			return raw
		out = torch.empty_like(raw)
		ops.rownorm_f32(raw, out)
		return out
