"""The reference's OpenCLIP and OpenAI embedder back-ends (embedders.py:438-594, :597-764) over LOCAL files -- this build never touches the network.

`Embedder.create('openclip:ORG/NAME')` / `'openai:ViT-B/32'` resolve the model against local storage only:

    1. NAME is itself a directory (or, for openai, a .pt file),
    2. $NOVIC_MODEL_ROOT/NAME                      (openai: $NOVIC_MODEL_ROOT/openai/<NAME with '/' -> '-'>.pt, or $OPENAI_HOME / ~/.cache/clip, where clip.load keeps it),
    3. the Hugging Face hub cache layout           $HF_HOME/hub/models--ORG--NAME/snapshots/<rev>/   (what open_clip's own 'hf-hub:' download leaves behind).

An OpenCLIP model directory is what the hub repository of the model holds (the reference loads `'hf-hub:' + model_id`, :614): `open_clip_config.json`
({"model_cfg": ..., "preprocess_cfg": ...}), `open_clip_pytorch_model.bin` or `open_clip_model.safetensors`, and the Hugging Face tokenizer files.  Host side as in the
reference: transformers' tokenizer classes (local_files_only), the special-token rules of :633-645 (bos/cls and eos/sep fallbacks, pad aliases, strip_sep_token), open_clip's
text cleaning modes; device side replaced by the native towers (clip_vit.NativeViT / clip_text.NativeTextTower, or the SigLIP variants of siglip.py for timm-trunk models),
which take open_clip's own state-dict names.
"""
from __future__ import annotations

import glob
import gzip
import html
import dataclasses
import json
import os
import re
import string
from typing import Any, Optional, Sequence, Union

import torch

from .embedders import Embedder


# ---- where models live -----------------------------------------------------------------------------------------------------------

def _hub_cache_dirs(model_id: str) -> list[str]:
	hf_home = os.path.expanduser(os.getenv("HF_HOME", os.path.join(os.getenv("XDG_CACHE_HOME", "~/.cache"), "huggingface")))
	hub = os.getenv("HF_HUB_CACHE", os.path.join(hf_home, "hub"))
	return sorted(glob.glob(os.path.join(hub, "models--" + model_id.replace("/", "--"), "snapshots", "*")), key=os.path.getmtime, reverse=True)


def resolve_model_dir(model_id: str, marker: str = "open_clip_config.json") -> str:
	"""The local directory of an open_clip hub model, or ValueError naming every place that was looked at."""
	tried = []
	cands = [model_id]
	root = os.getenv("NOVIC_MODEL_ROOT")
	if root:
		cands.append(os.path.join(os.path.expanduser(root), model_id))
	cands.extend(_hub_cache_dirs(model_id))
	for d in cands:
		tried.append(d)
		if os.path.isdir(d) and os.path.isfile(os.path.join(d, marker)):
			return d
	raise ValueError(f"OpenCLIP model '{model_id}' was not found on local storage (this build has no network access). Looked for {marker} in: {', '.join(tried)}. "
	                 f"Put the hub repository's files under $NOVIC_MODEL_ROOT/{model_id}/ (see INTEGRATION.md)")


def read_weights(model_dir: str, names: Sequence[str]) -> dict:
	for nm in names:
		path = os.path.join(model_dir, nm)
		if os.path.isfile(path):
			if nm.endswith(".safetensors"):
				from safetensors.torch import load_file
				return load_file(path)
			return torch.load(path, map_location="cpu", weights_only=True)
	raise ValueError(f"None of {list(names)} in {model_dir}")


# ---- open_clip's text cleaning modes (tokenizer.py of open_clip_torch 2.23: get_clean_fn) ------------------------------------------

def _basic_clean(text: str) -> str:
	try:
		import ftfy
		text = ftfy.fix_text(text)
	except ImportError:  # ftfy repairs mojibake; the vocabularies / prompts of this path are plain text, for which it is the identity
		pass
	return html.unescape(html.unescape(text)).strip()


def _whitespace_clean(text: str) -> str:
	return re.sub(r"\s+", " ", text).strip()


def _canonicalize(text: str) -> str:
	text = text.replace("_", " ").translate(str.maketrans("", "", string.punctuation)).lower()
	return re.sub(r"\s+", " ", text).strip()


CLEAN_FNS = {"whitespace": lambda x: _whitespace_clean(_basic_clean(x)), "lower": lambda x: _whitespace_clean(_basic_clean(x)).lower(),
             "canonicalize": lambda x: _canonicalize(_basic_clean(x))}


class OpenCLIPEmbedder(Embedder):
	"""reference embedders.py:597-764, model files from local storage (module docstring)."""

	def __init__(self, model_id: str, amp: bool = True, amp_bf16: bool = False, tokenizer_batch_size: int = 1024, inference_batch_size: int = 256, image_batch_size: int = 128,
	             load_model: bool = True, compile_model: bool = False, device: Union[int, str, torch.device] = "cuda", check: bool = False):
		import transformers
		self.model_id = model_id
		self.model_dir = resolve_model_dir(model_id)
		with open(os.path.join(self.model_dir, "open_clip_config.json"), "r", encoding="utf-8") as f:
			self.config = json.load(f)  # what open_clip.factory._get_hf_config returns (:617): the configuration hash below covers the same dict
		mc = self.config["model_cfg"]
		text_cfg = mc.get("text_cfg", {})
		tok_kw = dict(text_cfg.get("tokenizer_kwargs", {}))
		# open_clip.tokenizer.HFTokenizer(model_id, context_length, clean='whitespace', strip_sep_token=False) (:673-678) = AutoTokenizer of the repository's own files
		self.tokenizer = transformers.AutoTokenizer.from_pretrained(self.model_dir, local_files_only=True)
		self.tokenizer_context_length = int(text_cfg.get("context_length", 77))
		self.strip_sep_token = bool(tok_kw.get("strip_sep_token", False))
		clean = tok_kw.get("clean", "whitespace")
		if clean not in CLEAN_FNS:
			raise ValueError(f"Unknown open_clip text cleaning mode: {clean}")
		self.clean_fn = CLEAN_FNS[clean]
		self.tokenizer_clean = clean != "whitespace"  # :630: vocabulary / prompts are whitespace-perfect already, every other mode is applied
		tk = self.tokenizer
		start_token_id = tk.bos_token_id if tk.bos_token_id is not None else tk.cls_token_id
		end_token_id = tk.eos_token_id if tk.eos_token_id is not None else tk.sep_token_id
		end_token = tk.eos_token if tk.eos_token_id is not None else tk.sep_token
		pad_token_id, pad_token = tk.pad_token_id, tk.pad_token
		pad_aliases = {token for token, token_id in tk.get_vocab().items() if token_id == pad_token_id}
		pad_aliases.discard(pad_token)
		pad_aliases.discard(end_token)
		if pad_aliases:
			raise ValueError(f"Pad token {pad_token_id} cannot have non-end token aliases: {pad_aliases}")
		if self.strip_sep_token:
			end_token_id = pad_token_id
		self.text_tower = None
		super().__init__(configuration={"model_id": self.model_id, "model_config": self.config}, context_length=self.tokenizer_context_length, vocab_size=len(tk),
		                 cased_tokens=(tk.encode("CPU") != tk.encode("cpu")), start_token_id=start_token_id, end_token_id=end_token_id, pad_token_id=pad_token_id,
		                 token_dtype=torch.int64, embed_dtype=torch.float32, embed_dim=int(mc["embed_dim"]), amp_mode=False if not amp else torch.bfloat16 if amp_bf16 else True,
		                 manual_amp_dtype=None, tokenizer_batch_size=tokenizer_batch_size, inference_batch_size=inference_batch_size, image_batch_size=image_batch_size,
		                 load_model=load_model, compile_model=compile_model, device=device, check=check)

	# ---- model (reference :680-703) ----
	def load_model(self) -> bool:
		if self.is_model_loaded():
			return False
		sd = {k: v.float() for k, v in read_weights(self.model_dir, ("open_clip_model.safetensors", "open_clip_pytorch_model.bin", "open_clip_pytorch_model.pt")).items()
		      if torch.is_tensor(v) and v.is_floating_point()}
		vit, txt = build_towers(self.config["model_cfg"], sd, eot_from_argmax=True)
		if not txt.cfg.causal and self.pad_token_id is not None:
			# A tower without a causal mask sees its padding, so the rows must be padded to the context length with what open_clip's tokenizer call pads with: the
			# TOKENIZER's pad id (reference :735-738), not a `pad_id` the model config may or may not carry (open_clip's SigLIP configs have none; their pad id is 1 = '</s>')
			txt.cfg = dataclasses.replace(txt.cfg, pad_id=int(self.pad_token_id))
		vit.preprocess = self.config.get("preprocess_cfg", {})
		self.image_tower, self.text_tower = vit.to(self.device), txt.to(self.device)
		return True

	def unload_model(self) -> bool:
		if not self.is_model_loaded():
			return False
		self.image_tower = self.text_tower = None
		return True

	def is_model_loaded(self) -> bool:
		return self.image_tower is not None and self.text_tower is not None

	# ---- tokenizer (reference :705-726) ----
	def tokenize(self, text, max_tokens: Optional[int] = None, output_dict: bool = False):
		if max_tokens is None:
			max_tokens = self.context_length
		if self.tokenizer_clean:
			text = self.clean_fn(text) if isinstance(text, str) else tuple(self.clean_fn(t) for t in text)
		out = self.tokenizer(text=text, padding=True, truncation=True, max_length=max_tokens, return_tensors="pt")
		if self.strip_sep_token:
			ids = out.data["input_ids"]
			out.data["input_ids"] = torch.where(ids == self.tokenizer.sep_token_id, torch.tensor(self.pad_token_id, dtype=ids.dtype), ids)
		if not output_dict:
			return out["input_ids"]
		d = dict(out)
		if self.check:
			d["text"] = text
		return d

	def detokenize(self, token_ids: torch.Tensor):
		if token_ids.ndim <= 1:
			return self.tokenizer.decode(token_ids, skip_special_tokens=True)
		return self.tokenizer.batch_decode(token_ids, skip_special_tokens=True)

	def inference_tokens(self, tokens_dict: dict) -> torch.Tensor:
		if self.check and "text" in tokens_dict:  # :742-749: the ids must be what open_clip's own tokenizer call (pad to the context length) gives
			texts = tokens_dict["text"]
			chk = self.tokenizer(text=[texts] if isinstance(texts, str) else list(texts), return_tensors="pt", max_length=self.context_length, padding="max_length", truncation=True)["input_ids"]
			if self.strip_sep_token:
				chk = torch.where(chk == self.tokenizer.sep_token_id, torch.zeros_like(chk), chk)
			ids = tokens_dict["input_ids"].cpu()
			full = ids.new_full((ids.shape[0], self.context_length), self.pad_token_id)
			full[:, :ids.shape[1]] = ids
			if chk.dtype != full.dtype:
				raise ValueError("Token ID consistency check failed due to dtype")
			if not torch.equal(chk, full):
				raise ValueError("Token ID consistency check failed due to shape or value")
		return super().inference_tokens(tokens_dict)


def build_towers(mc: dict, sd: dict, eot_from_argmax: bool):
	"""open_clip `model_cfg` + state dict (open_clip names) -> (image tower, text tower) on the native kernels."""
	from . import clip_text, clip_vit
	vc, tc, F = mc["vision_cfg"], mc.get("text_cfg", {}), int(mc["embed_dim"])
	quick = bool(mc.get("quick_gelu", False))
	if "timm_model_name" in vc or any(k.startswith("visual.trunk.") for k in sd):
		from . import siglip
		return siglip.build_towers(mc, sd)
	for key in ("attentional_pool", "final_ln_after_pool", "pos_embed_type", "no_ln_pre", "ls_init_value", "patch_dropout", "output_tokens"):
		if vc.get(key) not in (None, False, 0, 0.0, "learnable"):
			raise NotImplementedError(f"open_clip vision_cfg.{key} = {vc[key]!r} is not implemented by the native image tower")
	if vc.get("pool_type", "cls") not in ("cls", "tok"):
		raise NotImplementedError(f"open_clip vision_cfg.pool_type = {vc['pool_type']!r} is not implemented by the native image tower")
	if tc.get("hf_model_name") or tc.get("no_causal_mask") or tc.get("embed_cls") or tc.get("proj_bias"):
		raise NotImplementedError("open_clip text_cfg with a Hugging Face text model / no causal mask / CLS embedding / projection bias is not implemented by the native text tower")
	W = int(vc["width"])
	hw = int(vc.get("head_width", 64))
	img = int(vc["image_size"]) if not isinstance(vc["image_size"], (list, tuple)) else int(vc["image_size"][0])
	vit = clip_vit.NativeViT(clip_vit.ViTConfig(image_size=img, patch_size=int(vc["patch_size"]), width=W, layers=int(vc["layers"]), heads=W // hw,
	                                            mlp_ratio=float(vc.get("mlp_ratio", 4.0)), embed_dim=F, quick_gelu=quick))
	vit.load_state_dict({k: v for k, v in sd.items() if k.startswith("visual.")})
	tw = int(tc.get("width", 512))
	txt = clip_text.NativeTextTower(clip_text.TextConfig(vocab_size=int(tc.get("vocab_size", 49408)), context_length=int(tc.get("context_length", 77)), width=tw,
	                                                     layers=int(tc.get("layers", 12)), heads=int(tc.get("heads", 8)), mlp_ratio=float(tc.get("mlp_ratio", 4.0)), embed_dim=F,
	                                                     quick_gelu=quick), eot_token_id=None if eot_from_argmax else int(tc.get("vocab_size", 49408)) - 1)
	text_keys = ("token_embedding.", "positional_embedding", "transformer.", "ln_final.", "text_projection")
	tsd = {(k[5:] if k.startswith("text.") else k): v for k, v in sd.items() if (k[5:] if k.startswith("text.") else k).startswith(text_keys) and not k.startswith("visual.")}
	txt.load_state_dict(tsd)
	return vit, txt


# ---- OpenAI CLIP ------------------------------------------------------------------------------------------------------------------

def _bytes_to_unicode() -> dict:
	"""The byte <-> printable-unicode table of byte-level BPE (GPT-2 / CLIP `simple_tokenizer.py`): printable bytes map to themselves, the others to 256 + n."""
	bs = list(range(ord("!"), ord("~") + 1)) + list(range(ord("\xa1"), ord("\xac") + 1)) + list(range(ord("\xae"), ord("\xff") + 1))
	cs = bs[:]
	n = 0
	for b in range(256):
		if b not in bs:
			bs.append(b)
			cs.append(256 + n)
			n += 1
	return dict(zip(bs, (chr(c) for c in cs)))


class SimpleBPE:
	"""CLIP's `SimpleTokenizer` (github.com/openai/CLIP clip/simple_tokenizer.py, the tokenizer `clip.clip._tokenizer` of reference :469), restated from its published
	algorithm: lower-cased, whitespace-cleaned text is split by the CLIP pattern, each piece is mapped to byte symbols with '</w>' on the last one, and adjacent pairs are
	merged in the rank order of the merges list.  Vocabulary layout: 256 byte symbols, the same 256 with '</w>', one entry per merge, then <|startoftext|>, <|endoftext|>.
	`merges`: the pair list (CLIP's bpe_simple_vocab_16e6.txt.gz holds it as lines 1 .. 49152-256-2 behind a header line)."""

	def __init__(self, merges: Sequence[tuple]):
		self.byte_encoder = _bytes_to_unicode()
		self.byte_decoder = {v: k for k, v in self.byte_encoder.items()}
		vocab = list(self.byte_encoder.values())
		vocab = vocab + [v + "</w>" for v in vocab]
		vocab.extend("".join(m) for m in merges)
		vocab.extend(["<|startoftext|>", "<|endoftext|>"])
		self.encoder = dict(zip(vocab, range(len(vocab))))
		self.decoder = {v: k for k, v in self.encoder.items()}
		self.bpe_ranks = dict(zip(merges, range(len(merges))))
		self.cache = {"<|startoftext|>": "<|startoftext|>", "<|endoftext|>": "<|endoftext|>"}
		try:
			import regex
			self.pat = regex.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""", regex.IGNORECASE)
		except ImportError:  # the unicode classes spelt with re's vocabulary
			self.pat = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[^\W\d_]+|\d|[^\s\w]+|_+""", re.IGNORECASE)

	@classmethod
	def from_file(cls, path: str) -> "SimpleBPE":
		opener = gzip.open if path.endswith(".gz") else open
		with opener(path, "rt", encoding="utf-8") as f:
			lines = f.read().split("\n")
		lines = lines[1:49152 - 256 - 2 + 1] if path.endswith(".gz") else [ln for ln in lines[1:] if ln]  # (merges.txt: '#version' header, one pair per line)
		return cls([tuple(ln.split()) for ln in lines if ln])

	def bpe(self, token: str) -> str:
		if token in self.cache:
			return self.cache[token]
		word = tuple(token[:-1]) + (token[-1] + "</w>",)
		while len(word) > 1:
			pairs = set(zip(word[:-1], word[1:]))
			bigram = min(pairs, key=lambda p: self.bpe_ranks.get(p, float("inf")))
			if bigram not in self.bpe_ranks:
				break
			first, second = bigram
			new, i = [], 0
			while i < len(word):
				if i < len(word) - 1 and word[i] == first and word[i + 1] == second:
					new.append(first + second)
					i += 2
				else:
					new.append(word[i])
					i += 1
			word = tuple(new)
		out = " ".join(word)
		self.cache[token] = out
		return out

	def encode(self, text: str) -> list:
		text = _whitespace_clean(_basic_clean(text)).lower()
		ids = []
		for piece in self.pat.findall(text):
			piece = "".join(self.byte_encoder[b] for b in piece.encode("utf-8"))
			ids.extend(self.encoder[t] for t in self.bpe(piece).split(" "))
		return ids

	def decode(self, tokens) -> str:
		text = "".join(self.decoder[int(t)] for t in tokens)
		return bytearray(self.byte_decoder[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


class OpenAIEmbedder(Embedder):
	"""reference embedders.py:438-594 over a local copy of what `clip.load` downloads: the model file (<name>.pt, TorchScript archive or state dict) and CLIP's BPE vocabulary
	(`bpe_simple_vocab_16e6.txt.gz`, shipped inside the clip package, or a Hugging Face `merges.txt` of the same vocabulary) next to it."""

	CONTEXT_LENGTH = 77
	EMBED_DIM = {"RN50": 1024, "RN101": 512, "RN50x4": 640, "RN50x16": 768, "RN50x64": 1024, "ViT-B/32": 512, "ViT-B/16": 512, "ViT-L/14": 768, "ViT-L/14@336px": 768}
	# clip.clip._MODELS: the URL is part of the reference's embedder configuration (:478), hence of the cache-file configuration hash
	MODELS = {
		"RN50": "https://openaipublic.azureedge.net/clip/models/afeb0e10f9e5a86da6080e35cf09123aca3b358a0c3e3b6c78a7b63bc04b6762/RN50.pt",
		"RN101": "https://openaipublic.azureedge.net/clip/models/8fa8567bab74a42d41c5915025a8e4538c3bdbe8804a470a72f30b0d94fab599/RN101.pt",
		"RN50x4": "https://openaipublic.azureedge.net/clip/models/7e526bd135e493cef0776de27d5f42653e6b4c8bf9e0f653bb11773263205fdd/RN50x4.pt",
		"RN50x16": "https://openaipublic.azureedge.net/clip/models/52378b407f34354e150460fe41077663dd5b39c54cd0bfd2b27167a4a06ec9aa/RN50x16.pt",
		"RN50x64": "https://openaipublic.azureedge.net/clip/models/be1cfb55d75a9666199fb2206c106743da0f6468c9d327f3e0d0a543a9919d9c/RN50x64.pt",
		"ViT-B/32": "https://openaipublic.azureedge.net/clip/models/40d365715913c9da98579312b702a82c18be219cc2a73407c4526f58eba950af/ViT-B-32.pt",
		"ViT-B/16": "https://openaipublic.azureedge.net/clip/models/5806e77cd80f8b59890b7e101eabd078d9fb84e6937f9e85e4ecb61988df416f/ViT-B-16.pt",
		"ViT-L/14": "https://openaipublic.azureedge.net/clip/models/b8cca3fd41ae0c99ba7e8951adf17d267cdb84cd88be6f7c2e0eca1737a03836/ViT-L-14.pt",
		"ViT-L/14@336px": "https://openaipublic.azureedge.net/clip/models/3035c92b350959924f9f00213499208652fc7ea050643e8b385c2dac08641f02/ViT-L-14-336px.pt",
	}

	def __init__(self, model_name: str, tokenizer_batch_size: int = 1024, inference_batch_size: int = 256, image_batch_size: int = 128, load_model: bool = True,
	             compile_model: bool = False, device: Union[int, str, torch.device] = "cuda", check: bool = False):
		if model_name not in self.MODELS:
			raise ValueError(f"Unknown OpenAI CLIP model: {model_name} (known: {sorted(self.MODELS)})")
		if model_name.startswith("RN"):
			raise NotImplementedError("The ResNet image towers of OpenAI CLIP are not implemented by the native kernels (ViT models only)")
		self.model_name = model_name
		self.model_path, vocab_path = self._resolve(model_name)
		self.tokenizer = SimpleBPE.from_file(vocab_path)
		self.text_tower = None
		enc = self.tokenizer.encoder
		super().__init__(configuration={"model_name": self.model_name, "model_checkpoint": self.MODELS[self.model_name]}, context_length=self.CONTEXT_LENGTH, vocab_size=len(enc),
		                 cased_tokens=False, start_token_id=enc["<|startoftext|>"], end_token_id=enc["<|endoftext|>"], pad_token_id=enc["<|endoftext|>"],  # (:484: pad = end)
		                 token_dtype=torch.int32, embed_dtype=torch.float32, embed_dim=self.EMBED_DIM[self.model_name], amp_mode=False, manual_amp_dtype=torch.float16,
		                 tokenizer_batch_size=tokenizer_batch_size, inference_batch_size=inference_batch_size, image_batch_size=image_batch_size, load_model=load_model,
		                 compile_model=compile_model, device=device, check=check)

	@classmethod
	def _resolve(cls, model_name: str) -> tuple:
		fname = os.path.basename(cls.MODELS[model_name])
		dirs = []
		root = os.getenv("NOVIC_MODEL_ROOT")
		if root:
			dirs.append(os.path.join(os.path.expanduser(root), "openai"))
		dirs.append(os.path.expanduser(os.getenv("OPENAI_HOME", os.path.join(os.getenv("XDG_CACHE_HOME", "~/.cache"), "clip"))))  # clip.load's download_root (:440)
		for d in dirs:
			model = os.path.join(d, fname)
			if os.path.isfile(model):
				for v in ("bpe_simple_vocab_16e6.txt.gz", "merges.txt"):
					if os.path.isfile(os.path.join(d, v)):
						return model, os.path.join(d, v)
				raise ValueError(f"Found {model} but no BPE vocabulary (bpe_simple_vocab_16e6.txt.gz or merges.txt) beside it")
		raise ValueError(f"OpenAI CLIP model '{model_name}' was not found on local storage (this build has no network access). Looked for {fname} in: {', '.join(dirs)}")

	def load_model(self) -> bool:
		if self.is_model_loaded():
			return False
		try:
			sd = torch.jit.load(self.model_path, map_location="cpu").state_dict()  # what clip.load reads: a TorchScript archive
		except RuntimeError:
			sd = torch.load(self.model_path, map_location="cpu", weights_only=True)
			sd = sd.get("state_dict", sd)
		sd = {k: v.float() for k, v in sd.items() if torch.is_tensor(v) and v.is_floating_point() and v.ndim >= 1}
		# clip/model.py build_model derives the architecture from the tensors
		W = sd["visual.conv1.weight"].shape[0]
		layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
		patch = sd["visual.conv1.weight"].shape[-1]
		grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
		tw = sd["ln_final.weight"].shape[0]
		mc = dict(embed_dim=sd["text_projection"].shape[1], quick_gelu=True,
		          vision_cfg=dict(image_size=patch * grid, patch_size=patch, width=W, layers=layers, head_width=64),
		          text_cfg=dict(context_length=sd["positional_embedding"].shape[0], vocab_size=sd["token_embedding.weight"].shape[0], width=tw, heads=tw // 64,
		                        layers=len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})))
		vit, txt = build_towers(mc, sd, eot_from_argmax=True)
		# what `clip.load` returns on a GPU is clip's HALF-PRECISION model ("already in a manual mixed precision configuration", reference :488-489): both towers'
		# residual streams are IEEE half here too (clip_vit.NativeViT.half_stream, round 6) -- for this family only; open_clip / transformers towers keep the fp32 stream
		# their reference runs under autocast
		vit.half_stream = txt.half_stream = True
		self.image_tower, self.text_tower = vit.to(self.device), txt.to(self.device)
		return True

	def unload_model(self) -> bool:
		if not self.is_model_loaded():
			return False
		self.image_tower = self.text_tower = None
		return True

	def is_model_loaded(self) -> bool:
		return self.image_tower is not None and self.text_tower is not None

	def tokenize(self, text, max_tokens: Optional[int] = None, output_dict: bool = False):
		"""reference :524-548"""
		if max_tokens is None:
			max_tokens = self.context_length
		texts = (text,) if isinstance(text, str) else text
		rows = []
		for t in texts:
			ids = [self.start_token_id] + self.tokenizer.encode(t)
			if len(ids) >= max_tokens:
				del ids[max_tokens - 1:]
			ids.append(self.end_token_id)
			rows.append(torch.tensor(ids, dtype=self.token_dtype))
		token_ids = torch.nn.utils.rnn.pad_sequence(rows, batch_first=True, padding_value=self.pad_token_id)
		if not output_dict:
			return token_ids
		att = torch.empty(token_ids.shape, dtype=self.token_dtype)
		att[:, 0] = 1
		att[:, 1:] = torch.ne(token_ids[:, :-1], self.pad_token_id)  # pad = end: everything up to and including the first end token
		d = {"input_ids": token_ids, "attention_mask": att}
		if self.check:
			d["text"] = texts
		return d

	def detokenize(self, token_ids: torch.Tensor):
		"""reference :550-555"""
		skip = (self.start_token_id, self.end_token_id)
		one = lambda row: self.tokenizer.decode(t for t in row if t not in skip).rstrip()
		rows = token_ids.tolist()
		return one(rows) if token_ids.ndim <= 1 else [one(r) for r in rows]
