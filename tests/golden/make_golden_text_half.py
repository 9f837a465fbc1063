#!/usr/bin/env python3
"""Round 6: pins for the HALF-PRECISION form of the CLIP text tower -- what the reference runs for its 'openai:' embedders (embedders.py:488-489: clip's fp16 model).
`transformers.CLIPTextModelWithProjection`, built from an explicit local config with the seeded weights of oracle.text_oracle.init_state_dict (no fetch), is cast to
torch.float16 and run ON THE CPU; `oracle.text_oracle.encode_text_half` must reproduce it before anything is written.  The fixture holds transformers' half-precision
embeddings and its fp32 ones for the same token rows.  Run in the build container:  python tests/golden/make_golden_text_half.py"""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from make_golden_text import hf_model, token_batch  # noqa: E402
from oracle import text_oracle as TO  # noqa: E402

CASES = [
	("tiny_quick_short", TO.TextSpec(vocab_size=500, context_length=24, width=128, layers=2, heads=2, embed_dim=32, quick_gelu=True), 4, 10),
	("b32_depth2_ctx77", TO.TextSpec(vocab_size=49408, context_length=77, width=512, layers=2, heads=8, embed_dim=512, quick_gelu=True), 3, 77),
	("b32_full_ctx77", TO.TextSpec(vocab_size=49408, context_length=77, width=512, layers=12, heads=8, embed_dim=512, quick_gelu=True), 4, 77),  # the text side of openai:ViT-B/32, all 12 layers
]


def main():
	out = []
	torch.set_num_threads(8)
	n = lambda t: torch.nn.functional.normalize(t.float(), dim=-1)
	for idx, (name, spec, B, S) in enumerate(CASES):
		seed = 950 + idx
		sd = TO.init_state_dict(spec, seed)
		ids = token_batch(spec, B, S, seed)
		with torch.no_grad():
			m = hf_model(spec, sd)
			r32 = m(input_ids=ids).text_embeds
			r16 = m.to(torch.float16)(input_ids=ids).text_embeds.float()
			mine = TO.encode_text_half(sd, spec, ids, normalize=False)
		cos_oracle = float((n(mine) * n(r16)).sum(-1).min())
		cos_prec = float((n(r32) * n(r16)).sum(-1).min())
		err = float((mine - r16).abs().max()) / max(1.0, float(r16.abs().max()))
		print(f"{name}: oracle(half) vs transformers(half): min cos {cos_oracle:.6f}, max |d| / scale {err:.2e};  transformers half vs fp32: min cos {cos_prec:.6f}")
		assert cos_oracle >= 0.99995 and err <= 4e-3, (name, cos_oracle, err)
		out.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, token_ids=ids, embeds_half_raw=r16.clone(), embeds_half=n(r16), embeds_fp32=n(r32), cos_half_vs_fp32=cos_prec))
	path = os.path.join(HERE, "text_forward_half.pt")
	torch.save(out, path)
	print(f"wrote text_forward_half.pt: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
	main()
