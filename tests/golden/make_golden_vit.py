#!/usr/bin/env python3
"""Golden vectors for the CLIP ViT image tower from `transformers.CLIPVisionModelWithProjection` built from an EXPLICIT LOCAL CONFIG
(random init, no fetch) -- the only stand-in for the reference's third-party `encode_image` available offline.
Weights are not stored: oracle.vit_oracle.init_state_dict(spec, seed) is loaded into the HF model through the key map below.
Run in the build container:  python tests/golden/make_golden_vit.py
"""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import vit_oracle as VO  # noqa: E402
from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection  # noqa: E402


def hf_model(spec: VO.ViTSpec, sd: dict):
	cfg = CLIPVisionConfig(hidden_size=spec.width, intermediate_size=spec.mlp_dim, projection_dim=spec.embed_dim, num_hidden_layers=spec.layers,
	                       num_attention_heads=spec.heads, image_size=spec.image_size, patch_size=spec.patch_size, hidden_act="quick_gelu" if spec.quick_gelu else "gelu",
	                       layer_norm_eps=spec.ln_eps, attention_dropout=0.0)
	m = CLIPVisionModelWithProjection(cfg).eval()
	W = spec.width
	hf = {
		"vision_model.embeddings.class_embedding": sd["visual.class_embedding"],
		"vision_model.embeddings.patch_embedding.weight": sd["visual.conv1.weight"],
		"vision_model.embeddings.position_embedding.weight": sd["visual.positional_embedding"],
		"vision_model.pre_layrnorm.weight": sd["visual.ln_pre.weight"], "vision_model.pre_layrnorm.bias": sd["visual.ln_pre.bias"],
		"vision_model.post_layernorm.weight": sd["visual.ln_post.weight"], "vision_model.post_layernorm.bias": sd["visual.ln_post.bias"],
		"visual_projection.weight": sd["visual.proj"].T.contiguous(),
	}
	for i in range(spec.layers):
		o, h = f"visual.transformer.resblocks.{i}.", f"vision_model.encoder.layers.{i}."
		for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
			hf[h + f"self_attn.{nm}.weight"] = sd[o + "attn.in_proj_weight"][j * W:(j + 1) * W]
			hf[h + f"self_attn.{nm}.bias"] = sd[o + "attn.in_proj_bias"][j * W:(j + 1) * W]
		hf[h + "self_attn.out_proj.weight"], hf[h + "self_attn.out_proj.bias"] = sd[o + "attn.out_proj.weight"], sd[o + "attn.out_proj.bias"]
		hf[h + "layer_norm1.weight"], hf[h + "layer_norm1.bias"] = sd[o + "ln_1.weight"], sd[o + "ln_1.bias"]
		hf[h + "layer_norm2.weight"], hf[h + "layer_norm2.bias"] = sd[o + "ln_2.weight"], sd[o + "ln_2.bias"]
		hf[h + "mlp.fc1.weight"], hf[h + "mlp.fc1.bias"] = sd[o + "mlp.c_fc.weight"], sd[o + "mlp.c_fc.bias"]
		hf[h + "mlp.fc2.weight"], hf[h + "mlp.fc2.bias"] = sd[o + "mlp.c_proj.weight"], sd[o + "mlp.c_proj.bias"]
	missing, unexpected = m.load_state_dict(hf, strict=False)
	assert not unexpected and all("position_ids" in k for k in missing), (missing, unexpected)
	return m


CASES = [
	("tiny_gelu", VO.ViTSpec(image_size=64, patch_size=16, width=128, layers=2, heads=4, embed_dim=64, quick_gelu=False), 3),
	("tiny_quick", VO.ViTSpec(image_size=96, patch_size=32, width=128, layers=2, heads=2, embed_dim=32, quick_gelu=True), 4),
	("p14_odd_k", VO.ViTSpec(image_size=56, patch_size=14, width=160, layers=1, heads=2, embed_dim=48, quick_gelu=False), 2),   # head_dim 80 (ViT-H), K = 588 (not /8)
	("b32_depth2", VO.ViTSpec(image_size=224, patch_size=32, width=768, layers=2, heads=12, embed_dim=512, quick_gelu=True), 2),  # ViT-B/32 dims, 2 of 12 layers
	("l14_depth1", VO.ViTSpec(image_size=224, patch_size=14, width=1024, layers=1, heads=16, embed_dim=768, quick_gelu=False), 1),  # ViT-L/14 dims (257 tokens), 1 layer
	("h14_depth1", VO.ViTSpec(image_size=224, patch_size=14, width=1280, layers=1, heads=16, embed_dim=1024, quick_gelu=False), 1),  # ViT-H/14 dims (configs[4]): head_dim 80, MLP 5120, F = 1024
]


# Full-depth / bench-shape pins (round 3).  Images are NOT stored (a seeded generator reproduces them, as it does the weights): the fixture holds transformers'
# embeddings of the first images of the seeded batch, and tests/test_gpu_vit.py runs the whole bench batch through NativeViT with the oracle tower -- pinned to
# transformers by these cases at the same depth -- as the checker.
FULL_CASES = [
	("b32_full", VO.ViTSpec(image_size=224, patch_size=32, width=768, layers=12, heads=12, embed_dim=512, quick_gelu=True), 4),  # ViT-B/32, all 12 layers (the metric's tower)
	("l14_depth2", VO.ViTSpec(image_size=224, patch_size=14, width=1024, layers=2, heads=16, embed_dim=768, quick_gelu=False), 2),  # ViT-L/14 dims, 2 of 24 layers
	# round 5: the towers configs[3] / configs[4] name at their FULL depth (bf16 error growth with depth was pinned for the 12-layer ViT-B/32 only)
	("l14_full", VO.ViTSpec(image_size=224, patch_size=14, width=1024, layers=24, heads=16, embed_dim=768, quick_gelu=False), 3),   # OpenCLIP ViT-L/14: all 24 layers, 257 tokens
	("h14_full", VO.ViTSpec(image_size=224, patch_size=14, width=1280, layers=32, heads=16, embed_dim=1024, quick_gelu=False), 2),  # OpenCLIP ViT-H/14: all 32 layers, head_dim 80, MLP 5120
]


def full_images(spec, seed, B):
	"""The seeded image batch of a full case: image i is the same for every B >= i + 1 (one generator call per image)."""
	g = torch.Generator().manual_seed(seed)
	return torch.stack([torch.randn(3, spec.image_size, spec.image_size, generator=g) for _ in range(B)])


def main_full():
	out = []
	for idx, (name, spec, B) in enumerate(FULL_CASES):
		seed = 700 + idx
		sd = VO.init_state_dict(spec, seed)
		images = full_images(spec, seed, B)
		with torch.no_grad():
			res = hf_model(spec, sd)(pixel_values=images)
			ref = res.image_embeds if hasattr(res, "image_embeds") else res.pooler_output
			mine = VO.encode_image(sd, spec, images, normalize=False)
		err = float((ref - mine).abs().max())
		assert err <= 2e-4 * max(1.0, float(ref.abs().max())), (name, err)
		out.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, batch=B, image_checksum=float(images.double().sum()), embeds_raw=ref.clone(),
		                embeds=torch.nn.functional.normalize(ref.float(), dim=-1)))
		print(name, "max |oracle - HF| =", err)
	path = os.path.join(HERE, "vit_forward_full.pt")
	torch.save(out, path)
	print(f"wrote vit_forward_full.pt: {os.path.getsize(path) / 1024:.1f} KiB")


def main():
	out = []
	for idx, (name, spec, B) in enumerate(CASES):
		seed = 500 + idx
		sd = VO.init_state_dict(spec, seed)
		g = torch.Generator().manual_seed(seed)
		images = torch.randn(B, 3, spec.image_size, spec.image_size, generator=g)
		with torch.no_grad():
			res = hf_model(spec, sd)(pixel_values=images)
			ref = res.image_embeds if hasattr(res, "image_embeds") else res.pooler_output
			mine = VO.encode_image(sd, spec, images, normalize=False)
		err = float((ref - mine).abs().max())
		assert err <= 2e-4 * max(1.0, float(ref.abs().max())), (name, err)
		out.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, images=images, embeds_raw=ref.clone(), embeds=torch.nn.functional.normalize(ref.float(), dim=-1)))
		print(name, "max |oracle - HF| =", err)
	path = os.path.join(HERE, "vit_forward.pt")
	torch.save(out, path)
	print(f"wrote vit_forward.pt: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
	if len(sys.argv) > 1 and sys.argv[1] == "full":
		main_full()
	else:
		main()
		main_full()
