#!/usr/bin/env python3
"""SigLIP in the open_clip hub-repository layout (what `openclip:timm/ViT-B-16-SigLIP` of two released checkpoints resolves to, reference README.md:293-298):

  tests/golden/openclip_tiny/testorg/ViT-tiny-SigLIP/   open_clip_config.json (timm trunk: timm_pool 'map', timm_proj 'none'; text: no_causal_mask, pool_type 'last',
                                                          proj_bias, tokenizer_kwargs.clean 'canonicalize'; preprocess 0.5 / 0.5), open_clip_model.safetensors with
                                                          open_clip's key names (`visual.trunk.*` = timm's, `text.*`), a sentencepiece-style (Unigram) tokenizer whose pad
                                                          token is the END token `</s>` and which has no start token (the SigLIP tokenizer's conventions)
  tests/golden/siglip_expected.pt                        token ids (transformers' tokenizer) and the embeddings of `transformers.SiglipModel` loaded with the SAME weights
                                                          through the key map below (hidden_act = 'gelu': exact erf, what the timm release of the reference's environment uses)
  tests/golden/siglip_so_expected.pt                     the same for a model with ViT-SO400M-14-SigLIP's odd dimensions in small (README.md:294: width 1152 = 16 heads of 72,
                                                          MLP 4304 -- here 144 = 2 heads of 72, MLP 200) and the tanh GELU (act_kwargs.approximate 'tanh' = transformers'
                                                          'gelu_pytorch_tanh'); the weights are NOT stored: the test rebuilds the directory from the seeds (the oracle's init is seeded)

open_clip and timm are not installed here; the oracle restatement (oracle/siglip_oracle.py) is cross-checked against transformers before anything is written.
Run in the build container (CPU): python tests/golden/make_golden_siglip.py"""
import json
import os
import sys

import torch
import transformers
from safetensors.torch import save_file
from tokenizers import Tokenizer, decoders, models, pre_tokenizers, processors

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import siglip_oracle as SO  # noqa: E402

PIECES = ["<pad>", "</s>", "<unk>", "▁", "▁a", "▁an", "▁photo", "▁of", "▁the", "▁in", "▁cat", "▁dog", "s", "▁bird", "▁house", "▁star", "ling", "▁ant", "c", "a", "t", "o", "d", "g", "e",
          "h", "u", "i", "r", "b", "n", "p", "l"]


def tokenizer(context: int):
	tok = Tokenizer(models.Unigram([(p, 0.0 if i < 3 else -float(i)) for i, p in enumerate(PIECES)], unk_id=2))
	tok.pre_tokenizer = pre_tokenizers.Metaspace(replacement="▁", prepend_scheme="always")
	tok.decoder = decoders.Metaspace(replacement="▁", prepend_scheme="always")
	tok.post_processor = processors.TemplateProcessing(single="$A </s>", special_tokens=[("</s>", 1)])
	return transformers.PreTrainedTokenizerFast(tokenizer_object=tok, eos_token="</s>", pad_token="</s>", unk_token="<unk>", model_max_length=context)


def hf_model(vs: SO.SigLIPVisionSpec, ts: SO.SigLIPTextSpec, sd: dict):
	act_v, act_t = ("gelu_pytorch_tanh" if sp.gelu_tanh else "gelu" for sp in (vs, ts))
	cfg = transformers.SiglipConfig(
		text_config=dict(vocab_size=ts.vocab_size, hidden_size=ts.width, intermediate_size=ts.mlp_dim, num_hidden_layers=ts.layers, num_attention_heads=ts.heads,
		                 max_position_embeddings=ts.context_length, projection_size=ts.embed_dim, hidden_act=act_t, layer_norm_eps=ts.ln_eps, pad_token_id=1, bos_token_id=None,
		                 eos_token_id=1),
		vision_config=dict(hidden_size=vs.width, intermediate_size=vs.mlp_dim, num_hidden_layers=vs.layers, num_attention_heads=vs.heads, image_size=vs.image_size,
		                   patch_size=vs.patch_size, hidden_act=act_v, layer_norm_eps=vs.ln_eps))
	m = transformers.SiglipModel(cfg).eval()
	t, a = "visual.trunk.", "visual.trunk.attn_pool."
	W = vs.width
	hf = {"vision_model.embeddings.patch_embedding.weight": sd[t + "patch_embed.proj.weight"], "vision_model.embeddings.patch_embedding.bias": sd[t + "patch_embed.proj.bias"],
	      "vision_model.embeddings.position_embedding.weight": sd[t + "pos_embed"][0], "vision_model.post_layernorm.weight": sd[t + "norm.weight"],
	      "vision_model.post_layernorm.bias": sd[t + "norm.bias"], "vision_model.head.probe": sd[a + "latent"],
	      "vision_model.head.attention.in_proj_weight": torch.cat((sd[a + "q.weight"], sd[a + "kv.weight"]), dim=0),
	      "vision_model.head.attention.in_proj_bias": torch.cat((sd[a + "q.bias"], sd[a + "kv.bias"]), dim=0),
	      "vision_model.head.attention.out_proj.weight": sd[a + "proj.weight"], "vision_model.head.attention.out_proj.bias": sd[a + "proj.bias"],
	      "vision_model.head.layernorm.weight": sd[a + "norm.weight"], "vision_model.head.layernorm.bias": sd[a + "norm.bias"],
	      "vision_model.head.mlp.fc1.weight": sd[a + "mlp.fc1.weight"], "vision_model.head.mlp.fc1.bias": sd[a + "mlp.fc1.bias"],
	      "vision_model.head.mlp.fc2.weight": sd[a + "mlp.fc2.weight"], "vision_model.head.mlp.fc2.bias": sd[a + "mlp.fc2.bias"],
	      "text_model.embeddings.token_embedding.weight": sd["text.token_embedding.weight"], "text_model.embeddings.position_embedding.weight": sd["text.positional_embedding"],
	      "text_model.final_layer_norm.weight": sd["text.ln_final.weight"], "text_model.final_layer_norm.bias": sd["text.ln_final.bias"],
	      "text_model.head.weight": sd["text.text_projection.weight"], "text_model.head.bias": sd["text.text_projection.bias"]}
	for i in range(vs.layers):
		o, h = f"{t}blocks.{i}.", f"vision_model.encoder.layers.{i}."
		for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
			hf[h + f"self_attn.{nm}.weight"], hf[h + f"self_attn.{nm}.bias"] = sd[o + "attn.qkv.weight"][j * W:(j + 1) * W], sd[o + "attn.qkv.bias"][j * W:(j + 1) * W]
		for src, dst in (("attn.proj", "self_attn.out_proj"), ("norm1", "layer_norm1"), ("norm2", "layer_norm2"), ("mlp.fc1", "mlp.fc1"), ("mlp.fc2", "mlp.fc2")):
			hf[h + dst + ".weight"], hf[h + dst + ".bias"] = sd[o + src + ".weight"], sd[o + src + ".bias"]
	Wt = ts.width
	for i in range(ts.layers):
		o, h = f"text.transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
		for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
			hf[h + f"self_attn.{nm}.weight"], hf[h + f"self_attn.{nm}.bias"] = sd[o + "attn.in_proj_weight"][j * Wt:(j + 1) * Wt], sd[o + "attn.in_proj_bias"][j * Wt:(j + 1) * Wt]
		for src, dst in (("attn.out_proj", "self_attn.out_proj"), ("ln_1", "layer_norm1"), ("ln_2", "layer_norm2"), ("mlp.c_fc", "mlp.fc1"), ("mlp.c_proj", "mlp.fc2")):
			hf[h + dst + ".weight"], hf[h + dst + ".bias"] = sd[o + src + ".weight"], sd[o + src + ".bias"]
	missing, unexpected = m.load_state_dict(hf, strict=False)
	assert not unexpected and set(missing) <= {"logit_scale", "logit_bias"} | {k for k in missing if "position_ids" in k}, (missing, unexpected)
	return m


def so_config(vocab: int, ctx: int) -> dict:
	"""open_clip_config.json of the small SO400M-shaped model (tests/test_local_clip.py rebuilds the directory with it)."""
	return {"model_cfg": {"embed_dim": 144, "init_logit_bias": -10, "custom_text": True,
	                      "vision_cfg": {"image_size": 56, "timm_model_name": "vit_so400m_patch14_siglip_56", "timm_model_pretrained": False, "timm_pool": "map", "timm_proj": "none",
	                                     "heads": 2, "act_kwargs": {"approximate": "tanh"}},
	                      "text_cfg": {"context_length": ctx, "vocab_size": vocab, "hf_tokenizer_name": "testorg/ViT-so-tiny-SigLIP", "tokenizer_kwargs": {"clean": "canonicalize"},
	                                   "width": 144, "heads": 2, "layers": 1, "mlp_ratio": 1.39, "no_causal_mask": True, "proj_bias": True, "pool_type": "last",
	                                   "norm_kwargs": {"eps": 1e-6}, "pad_id": 1, "act_kwargs": {"approximate": "tanh"}}},
	        "preprocess_cfg": {"mean": [0.5, 0.5, 0.5], "std": [0.5, 0.5, 0.5], "interpolation": "bicubic", "resize_mode": "squash"}}


def case(model_id, tok, ctx, vs, ts, cfg, seeds, image_seed, out_name, write_dir):
	sd = SO.init_vision_state_dict(vs, seeds[0])
	sd.update(SO.init_text_state_dict(ts, seeds[1]))
	sd["logit_scale"], sd["logit_bias"] = torch.tensor(2.3), torch.tensor(-10.0)
	if write_dir:
		d = os.path.join(HERE, "openclip_tiny", *model_id.split("/"))
		os.makedirs(d, exist_ok=True)
		tok.save_pretrained(d)
		with open(os.path.join(d, "open_clip_config.json"), "w") as f:
			json.dump(cfg, f, indent=2)
		save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "open_clip_model.safetensors"))
	raw = ["A photo of the cat!", "dogs", "The_starling (bird) house", "an ant in the house of the dog"]
	clean = ["a photo of the cat", "dogs", "the starling bird house", "an ant in the house of the dog"]
	enc = tok(text=clean, padding=True, truncation=True, max_length=ctx, return_tensors="pt")
	full = tok(text=clean, padding="max_length", truncation=True, max_length=ctx, return_tensors="pt")["input_ids"]  # open_clip's tokenizer call: padded to the context length
	g = torch.Generator().manual_seed(image_seed)
	images = torch.randn(3, 3, vs.image_size, vs.image_size, generator=g)
	m = hf_model(vs, ts, sd)
	with torch.no_grad():
		img_ref = m.get_image_features(pixel_values=images)
		txt_ref = m.get_text_features(input_ids=full)
		img_ref = img_ref.pooler_output if hasattr(img_ref, "pooler_output") else img_ref
		txt_ref = txt_ref.pooler_output if hasattr(txt_ref, "pooler_output") else txt_ref
		img_mine, txt_mine = SO.encode_image(sd, vs, images, normalize=False), SO.encode_text(sd, ts, full, normalize=False)
	for nm, a, b in (("image", img_ref, img_mine), ("text", txt_ref, txt_mine)):
		err = float((a - b).abs().max())
		assert err <= 2e-4 * max(1.0, float(a.abs().max())), (nm, err)
		print(f"  {model_id} {nm}: max |oracle - transformers.SiglipModel| = {err:.2e}")
	torch.save(dict(model_id=model_id, texts=raw, clean=clean, input_ids=enc["input_ids"], attention_mask=enc["attention_mask"], input_ids_full=full,
	                images=images, image_embeds=torch.nn.functional.normalize(img_ref.float(), dim=-1), text_embeds=torch.nn.functional.normalize(txt_ref.float(), dim=-1),
	                image_embeds_raw=img_ref.float(), special=dict(start=None, end=tok.eos_token_id, pad=tok.pad_token_id, vocab=len(tok), context=ctx), config=cfg,
	                vision_spec=vs.__dict__, text_spec=ts.__dict__, seeds=seeds, transformers=transformers.__version__), os.path.join(HERE, out_name))


def main():
	ctx = 16
	tok = tokenizer(ctx)
	vs = SO.SigLIPVisionSpec(image_size=64, patch_size=16, width=64, layers=2, heads=1, mlp_dim=256)
	ts = SO.SigLIPTextSpec(vocab_size=len(tok), context_length=ctx, width=64, layers=2, heads=1, mlp_dim=256, embed_dim=64)
	cfg = {"model_cfg": {"embed_dim": 64, "init_logit_bias": -10, "custom_text": True,
	                     "vision_cfg": {"image_size": 64, "timm_model_name": "vit_tiny_patch16_siglip_64", "timm_model_pretrained": False, "timm_pool": "map", "timm_proj": "none", "heads": 1},
	                     "text_cfg": {"context_length": ctx, "vocab_size": len(tok), "hf_tokenizer_name": "testorg/ViT-tiny-SigLIP", "tokenizer_kwargs": {"clean": "canonicalize"},
	                                  "width": 64, "heads": 1, "layers": 2, "no_causal_mask": True, "proj_bias": True, "pool_type": "last", "norm_kwargs": {"eps": 1e-6}, "pad_id": 1}},
	       "preprocess_cfg": {"mean": [0.5, 0.5, 0.5], "std": [0.5, 0.5, 0.5], "interpolation": "bicubic", "resize_mode": "squash"}}
	case("testorg/ViT-tiny-SigLIP", tok, ctx, vs, ts, cfg, (61, 62), 63, "siglip_expected.pt", write_dir=True)
	# ViT-SO400M-14-SigLIP's dimensions in small: 72-wide heads, an MLP width that is no multiple of 64, the tanh GELU
	vs = SO.SigLIPVisionSpec(image_size=56, patch_size=14, width=144, layers=1, heads=2, mlp_dim=200, gelu_tanh=True)
	ts = SO.SigLIPTextSpec(vocab_size=len(tok), context_length=ctx, width=144, layers=1, heads=2, mlp_dim=int(144 * 1.39), embed_dim=144, gelu_tanh=True)
	assert ts.mlp_dim == 200
	case("testorg/ViT-so-tiny-SigLIP", tok, ctx, vs, ts, so_config(len(tok), ctx), (64, 65), 66, "siglip_so_expected.pt", write_dir=False)
	print("wrote openclip_tiny/testorg/ViT-tiny-SigLIP/, siglip_expected.pt, siglip_so_expected.pt")


if __name__ == "__main__":
	main()
