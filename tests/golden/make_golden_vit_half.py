#!/usr/bin/env python3
"""Round 6: pins for the HALF-PRECISION form of the CLIP image tower -- what the reference runs for its 'openai:' embedders (embedders.py:488-489: clip's fp16 model).
`transformers.CLIPVisionModelWithProjection`, built from an explicit local config with the seeded weights of oracle.vit_oracle.init_state_dict (no fetch), is cast to
torch.float16 and run ON THE CPU; `oracle.vit_oracle.encode_image_half` (every tensor between two operations rounded to half, fp32 accumulation and LayerNorm statistics)
must reproduce it before anything is written.  The fixture holds transformers' half-precision embeddings AND its fp32 ones for the same images, so that a test can see
how far the two precisions of the reference itself are apart.
Run in the build container:  python tests/golden/make_golden_vit_half.py"""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden_vit import hf_model, full_images  # noqa: E402
from oracle import vit_oracle as VO  # noqa: E402

CASES = [
	("tiny_quick", VO.ViTSpec(image_size=96, patch_size=32, width=128, layers=2, heads=2, embed_dim=32, quick_gelu=True), 4, 0),
	("b32_depth2", VO.ViTSpec(image_size=224, patch_size=32, width=768, layers=2, heads=12, embed_dim=512, quick_gelu=True), 2, 0),
	("b32_full", VO.ViTSpec(image_size=224, patch_size=32, width=768, layers=12, heads=12, embed_dim=512, quick_gelu=True), 4, 700),  # the metric's tower, all 12 layers: the images of vit_forward_full.pt's b32_full (seed 700)
]


def main():
	out = []
	torch.set_num_threads(8)
	for idx, (name, spec, B, seed0) in enumerate(CASES):
		seed = seed0 or 900 + idx
		sd = VO.init_state_dict(spec, seed)
		images = full_images(spec, seed, B)
		with torch.no_grad():
			m = hf_model(spec, sd)
			r32 = m(pixel_values=images)
			r32 = r32.image_embeds if hasattr(r32, "image_embeds") else r32.pooler_output
			mh = m.to(torch.float16)
			r16 = mh(pixel_values=images.to(torch.float16))
			r16 = (r16.image_embeds if hasattr(r16, "image_embeds") else r16.pooler_output).float()
			mine = VO.encode_image_half(sd, spec, images, normalize=False)
		n = lambda t: torch.nn.functional.normalize(t.float(), dim=-1)
		cos_oracle = float((n(mine) * n(r16)).sum(-1).min())
		cos_prec = float((n(r32) * n(r16)).sum(-1).min())
		err = float((mine - r16).abs().max()) / max(1.0, float(r16.abs().max()))
		print(f"{name}: oracle(half) vs transformers(half): min cos {cos_oracle:.6f}, max |d| / scale {err:.2e};  transformers half vs fp32: min cos {cos_prec:.6f}")
		assert cos_oracle >= 0.99995 and err <= 4e-3, (name, cos_oracle, err)  # two half-precision evaluations that differ in summation order: a few ulps of half per element
		out.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, batch=B, image_checksum=float(images.double().sum()), embeds_half_raw=r16.clone(), embeds_half=n(r16),
		                embeds_fp32=n(r32), cos_half_vs_fp32=cos_prec))
	path = os.path.join(HERE, "vit_forward_half.pt")
	torch.save(out, path)
	print(f"wrote vit_forward_half.pt: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
	main()
