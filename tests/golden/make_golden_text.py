#!/usr/bin/env python3
"""Golden vectors for the CLIP text tower from `transformers.CLIPTextModelWithProjection` built from an EXPLICIT LOCAL CONFIG (random init, no
fetch) -- the only stand-in for the reference's third-party `encode_text` available offline.  Weights are not stored:
oracle.text_oracle.init_state_dict(spec, seed) is loaded into the HF model through the key map below.
Run in the build container:  python tests/golden/make_golden_text.py
"""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import text_oracle as TO  # noqa: E402
from transformers import CLIPTextConfig, CLIPTextModelWithProjection  # noqa: E402


def hf_model(spec: TO.TextSpec, sd: dict):
	# eos_token_id = 2 selects the original CLIP pooling (row of the arg-max token id), which is what OpenAI CLIP / open_clip do
	cfg = CLIPTextConfig(vocab_size=spec.vocab_size, hidden_size=spec.width, intermediate_size=spec.mlp_dim, projection_dim=spec.embed_dim, num_hidden_layers=spec.layers,
	                     num_attention_heads=spec.heads, max_position_embeddings=spec.context_length, hidden_act="quick_gelu" if spec.quick_gelu else "gelu",
	                     layer_norm_eps=spec.ln_eps, attention_dropout=0.0, eos_token_id=2, bos_token_id=0, pad_token_id=1)
	m = CLIPTextModelWithProjection(cfg).eval()
	W = spec.width
	hf = {
		"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
		"text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
		"text_model.final_layer_norm.weight": sd["ln_final.weight"], "text_model.final_layer_norm.bias": sd["ln_final.bias"],
		"text_projection.weight": sd["text_projection"].T.contiguous(),
	}
	for i in range(spec.layers):
		o, h = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
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


def token_batch(spec, B, S, seed):
	"""CLIP-style rows: <start> content... <end = largest id> then padding with the end id's predecessor class (zeros), ragged lengths."""
	g = torch.Generator().manual_seed(seed)
	ids = torch.zeros(B, S, dtype=torch.int64)
	eot, sot = spec.vocab_size - 1, spec.vocab_size - 2
	for b in range(B):
		n = int(torch.randint(1, S - 1, (1,), generator=g))
		ids[b, 0] = sot
		ids[b, 1:1 + n] = torch.randint(1, spec.vocab_size - 2, (n,), generator=g)
		ids[b, 1 + n] = eot
	return ids


CASES = [
	("tiny_gelu", TO.TextSpec(vocab_size=300, context_length=16, width=128, layers=2, heads=4, embed_dim=64, quick_gelu=False), 5, 16),
	("tiny_quick_short", TO.TextSpec(vocab_size=500, context_length=24, width=128, layers=2, heads=2, embed_dim=32, quick_gelu=True), 4, 10),  # S < context length
	("b32_depth2_ctx77", TO.TextSpec(vocab_size=49408, context_length=77, width=512, layers=2, heads=8, embed_dim=512, quick_gelu=True), 3, 77),  # CLIP ViT-B/32 text dims, 2 of 12 layers
	("l14_depth1_ctx77", TO.TextSpec(vocab_size=49408, context_length=77, width=768, layers=1, heads=12, embed_dim=768, quick_gelu=False), 2, 77),  # ViT-L/14 text dims
]


def main():
	out = []
	for idx, (name, spec, B, S) in enumerate(CASES):
		seed = 800 + idx
		sd = TO.init_state_dict(spec, seed)
		ids = token_batch(spec, B, S, seed)
		with torch.no_grad():
			ref = hf_model(spec, sd)(input_ids=ids).text_embeds
			mine = TO.encode_text(sd, spec, ids, normalize=False)
		err = float((ref - mine).abs().max())
		assert err <= 2e-4 * max(1.0, float(ref.abs().max())), (name, err)
		out.append(dict(name=name, spec=dataclasses.asdict(spec), seed=seed, token_ids=ids, embeds_raw=ref.clone(), embeds=torch.nn.functional.normalize(ref.float(), dim=-1)))
		print(name, "max |oracle - HF| =", err)
	path = os.path.join(HERE, "text_forward.pt")
	torch.save(out, path)
	print(f"wrote text_forward.pt: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
	main()
