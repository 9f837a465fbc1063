#!/usr/bin/env python3
"""Writes local model files in the layouts the reference's OpenCLIP / OpenAI embedders fetch by name (embedders.py:438-764), and what they must produce:

  tests/golden/openclip_tiny/testorg/CLIP-ViT-tiny-quickgelu/   an open_clip hub repository: open_clip_config.json, open_clip_model.safetensors (open_clip's own
                                                                state-dict names), Hugging Face tokenizer files (synthetic CLIP-layout BPE vocabulary)
  tests/golden/openclip_tiny/testorg/CLIPA-tiny-bert/           the same towers behind a BERT-style tokenizer (cls / sep, no bos / eos) with `strip_sep_token`
                                                                and the 'canonicalize' cleaning mode: the special-token fallbacks of embedders.py:633-645
  tests/golden/openai_tiny/ViT-B-32.pt + merges.txt             what `clip.load('ViT-B/32')` keeps in its download directory, at toy dims (the architecture is
                                                                derived from the tensors, clip/model.py build_model) + CLIP's BPE merges
  tests/golden/openclip_tiny_expected.pt                        token ids (transformers' tokenizer), embeddings

open_clip and clip are not installed here.  Their CLIP architecture is the one transformers' CLIPModel implements, so the expected embeddings come from
transformers' CLIPTextModelWithProjection / CLIPVisionModelWithProjection loaded with the SAME weights through the key maps of make_golden_vit.py / make_golden_text.py,
cross-checked against the oracle towers before writing.  Run in the build container (CPU): python tests/golden/make_golden_openclip.py"""
import dataclasses
import json
import os
import sys

import torch
import transformers
from safetensors.torch import save_file

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
from oracle import text_oracle as TO, vit_oracle as VO  # noqa: E402
import make_golden_hfclip as HC  # noqa: E402
import make_golden_text as MT  # noqa: E402
import make_golden_vit as MV  # noqa: E402

CLIP_MEAN, CLIP_STD = (0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)
MERGES = [("t", "h"), ("th", "e</w>"), ("c", "a"), ("ca", "t</w>"), ("d", "o"), ("do", "g</w>"), ("i", "n"), ("in", "g</w>"), ("a", "n</w>"), ("o", "f</w>"), ("p", "h"),
          ("ph", "o"), ("pho", "t"), ("phot", "o</w>"), ("b", "i"), ("bi", "r"), ("bir", "d</w>"), ("s", "t"), ("st", "a"), ("sta", "r"), ("h", "o"), ("ho", "u"), ("hou", "s"),
          ("hous", "e</w>")]
TEXTS = ["a photo of the cat", "dog", "the starling bird house", "an ant in the house of the dog", "Photo of CAT"]


def clip_tokenizer():
	chars = list(HC.bytes_to_unicode().values())
	vocab = chars + [c + "</w>" for c in chars] + [a + b for a, b in MERGES] + ["<|startoftext|>", "<|endoftext|>"]
	return transformers.CLIPTokenizer(vocab={t: i for i, t in enumerate(vocab)}, merges=MERGES, model_max_length=77)


def towers(vocab_size: int, seed: int):
	vspec = VO.ViTSpec(image_size=64, patch_size=16, width=64, layers=2, heads=1, embed_dim=64, quick_gelu=True)   # head width 64 (OpenAI's rule: heads = width / 64)
	tspec = TO.TextSpec(vocab_size=vocab_size, context_length=77, width=64, layers=2, heads=1, embed_dim=64, quick_gelu=True)
	vsd, tsd = VO.init_state_dict(vspec, seed), TO.init_state_dict(tspec, seed + 1)
	sd = dict(vsd)
	sd.update(tsd)
	sd["logit_scale"] = torch.tensor(2.6592)
	return vspec, tspec, vsd, tsd, sd


def expected(vspec, tspec, vsd, tsd, ids, images):
	with torch.no_grad():
		res = MV.hf_model(vspec, vsd)(pixel_values=images)
		img_ref = res.image_embeds if hasattr(res, "image_embeds") else res.pooler_output
		txt_ref = MT.hf_model(tspec, tsd)(input_ids=ids).text_embeds
		img_mine, txt_mine = VO.encode_image(vsd, vspec, images, normalize=False), TO.encode_text(tsd, tspec, ids, normalize=False)
	for nm, a, b in (("image", img_ref, img_mine), ("text", txt_ref, txt_mine)):
		err = float((a - b).abs().max())
		assert err <= 2e-4 * max(1.0, float(a.abs().max())), (nm, err)
		print(f"    {nm}: max |oracle - transformers| = {err:.2e}")
	return torch.nn.functional.normalize(img_ref.float(), dim=-1), torch.nn.functional.normalize(txt_ref.float(), dim=-1)


def main():
	out = {}
	g = torch.Generator().manual_seed(21)
	images = torch.randn(3, 3, 64, 64, generator=g)

	# ---- 1. an open_clip hub repository with a CLIP BPE tokenizer ----
	d = os.path.join(HERE, "openclip_tiny", "testorg", "CLIP-ViT-tiny-quickgelu")
	os.makedirs(d, exist_ok=True)
	tok = clip_tokenizer()
	tok.save_pretrained(d)
	vspec, tspec, vsd, tsd, sd = towers(len(tok), 31)
	cfg = {"model_cfg": {"embed_dim": 64, "quick_gelu": True,
	                     "vision_cfg": {"image_size": 64, "layers": 2, "width": 64, "patch_size": 16},
	                     "text_cfg": {"context_length": 77, "vocab_size": len(tok), "width": 64, "heads": 1, "layers": 2}},
	       "preprocess_cfg": {"mean": list(CLIP_MEAN), "std": list(CLIP_STD)}}
	with open(os.path.join(d, "open_clip_config.json"), "w") as f:
		json.dump(cfg, f, indent=2)
	save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "open_clip_model.safetensors"))
	enc = tok(text=TEXTS, padding=True, truncation=True, max_length=77, return_tensors="pt")
	print("  open_clip layout, CLIP tokenizer:")
	img_e, txt_e = expected(vspec, tspec, vsd, tsd, enc["input_ids"], images)
	out["clip"] = dict(model_id="testorg/CLIP-ViT-tiny-quickgelu", texts=TEXTS, input_ids=enc["input_ids"], attention_mask=enc["attention_mask"],
	                   decoded=tok.batch_decode(enc["input_ids"], skip_special_tokens=True), images=images, image_embeds=img_e, text_embeds=txt_e,
	                   special=dict(start=tok.bos_token_id, end=tok.eos_token_id, pad=tok.pad_token_id, vocab=len(tok), context=77), config=cfg)

	# ---- 2. the same architecture behind a BERT-style tokenizer: cls / sep fallbacks, strip_sep_token, canonicalize ----
	d2 = os.path.join(HERE, "openclip_tiny", "testorg", "CLIPA-tiny-bert")
	os.makedirs(d2, exist_ok=True)
	words = ["[PAD]", "[UNK]", "[CLS]", "[SEP]", "[MASK]", "a", "an", "the", "of", "in", "photo", "cat", "dog", "bird", "house", "star", "##ling", "ant", "##s"]
	btok = transformers.BertTokenizer(vocab={w: i for i, w in enumerate(words)}, do_lower_case=True, model_max_length=32)
	assert btok("dogs")["input_ids"] == [2, 12, 18, 3]
	btok.save_pretrained(d2)
	vspec2, tspec2, vsd2, tsd2, sd2 = towers(len(btok), 41)
	tspec2 = dataclasses.replace(tspec2, context_length=32)
	tsd2 = TO.init_state_dict(tspec2, 42)
	sd2 = dict(vsd2)
	sd2.update(tsd2)
	cfg2 = {"model_cfg": {"embed_dim": 64, "quick_gelu": True,
	                      "vision_cfg": {"image_size": 64, "layers": 2, "width": 64, "patch_size": 16},
	                      "text_cfg": {"context_length": 32, "vocab_size": len(btok), "width": 64, "heads": 1, "layers": 2,
	                                   "tokenizer_kwargs": {"strip_sep_token": True, "clean": "canonicalize"}}},
	        "preprocess_cfg": {"mean": [0.5, 0.5, 0.5], "std": [0.5, 0.5, 0.5], "interpolation": "bicubic"}}
	with open(os.path.join(d2, "open_clip_config.json"), "w") as f:
		json.dump(cfg2, f, indent=2)
	torch.save({k: v.clone() for k, v in sd2.items()}, os.path.join(d2, "open_clip_pytorch_model.bin"))
	raw = ["A photo of the cat!", "dogs", "The_starling, (bird) house"]
	clean = ["a photo of the cat", "dogs", "the starling bird house"]  # open_clip canonicalize_text: '_' -> ' ', punctuation removed, lower case, whitespace collapsed
	enc2 = btok(text=clean, padding=True, truncation=True, max_length=32, return_tensors="pt")
	ids2 = torch.where(enc2["input_ids"] == btok.sep_token_id, torch.tensor(btok.pad_token_id), enc2["input_ids"])  # strip_sep_token (open_clip HFTokenizer.__call__)
	out["bert"] = dict(model_id="testorg/CLIPA-tiny-bert", texts=raw, clean=clean, input_ids=ids2, attention_mask=enc2["attention_mask"],
	                   special=dict(start=btok.cls_token_id, end=btok.pad_token_id, pad=btok.pad_token_id, vocab=len(btok), context=32), config=cfg2)

	# ---- 3. OpenAI CLIP: the downloaded .pt (here a plain state dict at toy dims) + the BPE merges ----
	d3 = os.path.join(HERE, "openai_tiny")
	os.makedirs(d3, exist_ok=True)
	with open(os.path.join(d3, "merges.txt"), "w") as f:
		f.write("#version: 0.2\n" + "\n".join(f"{a} {b}" for a, b in MERGES) + "\n")
	vspec3, tspec3, vsd3, tsd3, sd3 = towers(len(tok), 51)
	sd3.update(input_resolution=torch.tensor(64), context_length=torch.tensor(77), vocab_size=torch.tensor(len(tok)))
	torch.save({k: v.clone() for k, v in sd3.items()}, os.path.join(d3, "ViT-B-32.pt"))
	# CLIP's tokenize(): <start> + BPE + <end>, int32, padded with the END id (the reference's choice, :484) instead of clip.tokenize's zeros
	rows = [[tok.bos_token_id] + tok(t, add_special_tokens=False)["input_ids"] + [tok.eos_token_id] for t in TEXTS]
	L = max(len(r) for r in rows)
	ids3 = torch.full((len(rows), L), tok.eos_token_id, dtype=torch.int32)
	for i, r in enumerate(rows):
		ids3[i, :len(r)] = torch.tensor(r, dtype=torch.int32)
	print("  OpenAI layout:")
	img_e3, txt_e3 = expected(vspec3, tspec3, vsd3, tsd3, ids3.long(), images)
	out["openai"] = dict(model_name="ViT-B/32", texts=TEXTS, input_ids=ids3, decoded=[t.lower() for t in TEXTS], images=images, image_embeds=img_e3, text_embeds=txt_e3,
	                     special=dict(start=tok.bos_token_id, end=tok.eos_token_id, pad=tok.eos_token_id, vocab=len(tok), context=77))
	out["transformers"] = transformers.__version__
	torch.save(out, os.path.join(HERE, "openclip_tiny_expected.pt"))
	print("wrote openclip_tiny/, openai_tiny/, openclip_tiny_expected.pt")


if __name__ == "__main__":
	main()
