#!/usr/bin/env python3
"""Writes tests/golden/hf_clip_tiny/ -- a LOCAL Hugging Face CLIP directory (config.json, tokenizer files, model.safetensors) with a synthetic
byte-level BPE vocabulary and random-init weights -- and tests/golden/hf_clip_tiny_expected.pt: token ids and the unit-norm fp32 embeddings of
transformers' own CLIPModel.get_text_features / get_image_features (the calls the reference's TransformersEmbedder makes, embedders.py:890, :906)
on fixed texts / images.  Run in the build container (CPU): python tests/golden/make_golden_hfclip.py"""
import os

import torch
import transformers

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "hf_clip_tiny")


def bytes_to_unicode():
	bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
	cs, n = bs[:], 0
	for b in range(256):
		if b not in bs:
			bs.append(b)
			cs.append(256 + n)
			n += 1
	return dict(zip(bs, [chr(c) for c in cs]))


def main():
	os.makedirs(OUT, exist_ok=True)
	chars = list(bytes_to_unicode().values())
	vocab = chars + [c + "</w>" for c in chars]
	merges = [("t", "h"), ("th", "e</w>"), ("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>"), ("i", "n"), ("in", "g</w>"), ("a", "n</w>"), ("o", "f</w>"), ("p", "h"),
	          ("ph", "o"), ("pho", "t"), ("phot", "o</w>"), ("b", "i"), ("bi", "r"), ("bir", "d</w>"), ("s", "t"), ("st", "a"), ("sta", "r"), ("h", "o"), ("ho", "u"), ("hou", "s"),
	          ("hous", "e</w>")]
	vocab += [a + b for a, b in merges] + ["<|startoftext|>", "<|endoftext|>"]
	tok = transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(vocab)}, merges=merges, model_max_length=77)
	tok.save_pretrained(OUT)
	cfg = transformers.CLIPConfig(
		text_config=dict(vocab_size=len(tok), hidden_size=64, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, max_position_embeddings=77,
		                 eos_token_id=tok.eos_token_id, bos_token_id=tok.bos_token_id, pad_token_id=tok.pad_token_id, hidden_act="quick_gelu"),
		vision_config=dict(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4, image_size=64, patch_size=16, hidden_act="quick_gelu"),
		projection_dim=64)
	torch.manual_seed(0)
	model = transformers.CLIPModel(cfg).eval()
	model.save_pretrained(OUT)
	texts = ["a photo of the cat", "dog", "the starling bird house", "an ant in the house of the dog", "Photo of CAT"]
	enc = tok(text=texts, padding=True, truncation=True, max_length=None, return_tensors="pt")
	g = torch.Generator().manual_seed(1)
	images = torch.randn(3, 3, 64, 64, generator=g)
	with torch.no_grad():
		tf = model.get_text_features(**enc)
		imf = model.get_image_features(pixel_values=images)
	tf = tf.pooler_output if hasattr(tf, "pooler_output") else tf  # transformers 5: BaseModelOutputWithPooling, the projected embedding in pooler_output
	imf = imf.pooler_output if hasattr(imf, "pooler_output") else imf
	torch.save({"texts": texts, "input_ids": enc["input_ids"], "attention_mask": enc["attention_mask"], "decoded": tok.batch_decode(enc["input_ids"], skip_special_tokens=True),
	            "text_embeds": torch.nn.functional.normalize(tf.float(), dim=-1), "images": images, "image_embeds": torch.nn.functional.normalize(imf.float(), dim=-1),
	            "special": dict(bos=tok.bos_token_id, eos=tok.eos_token_id, pad=tok.pad_token_id, vocab=len(tok), context=tok.model_max_length),
	            "transformers": transformers.__version__}, os.path.join(HERE, "hf_clip_tiny_expected.pt"))
	print("wrote", OUT, os.listdir(OUT))


if __name__ == "__main__":
	main()
