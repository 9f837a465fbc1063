#!/usr/bin/env python3
"""Golden embedding-cache files + expected batches, produced by the reference's OWN writer (embedding_cache.EmbeddingCacheWriter) and reader
(EmbeddingCache.Dataset.__getitem__).  Build container only:  python tests/golden/make_golden_cache.py
The .bin fixtures are data files in the reference's cache format; cache_batches.pt holds the batches the reference reads back from them."""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.modules.setdefault("unidecode", types.ModuleType("unidecode"))
sys.modules["unidecode"].unidecode = lambda s: s

import embedders as ref_embedders  # noqa: E402
import embedding_cache as ref_cache  # noqa: E402
import embedding_dataset as ref_dataset  # noqa: E402

TOKENS = ["red", "panda", "fire", "truck", "sea", "lion", "ice", "cream", "cone", "dog", "cat", "house"]
NOUNS = ("red panda", "fire truck", "sea lion", "ice cream cone", "dog", "cat", "house", "dog house", "sea")
F = 16


class LocalVocabEmbedder(ref_embedders.Embedder):
	"""Reference-side twin of novic_amd.embedders.LocalVocabEmbedder (same class name + configuration => same hashes)."""

	def __init__(self):
		self.itos = ["<pad>", "<start>", "<end>"] + TOKENS
		self.stoi = {t: i for i, t in enumerate(self.itos)}
		super().__init__(configuration=dict(type="local", num_tokens=len(TOKENS), embed_dim=F, context_length=77, with_start=True), context_length=77, vocab_size=len(self.itos),
		                 cased_tokens=True, start_token_id=1, end_token_id=2, pad_token_id=0, token_dtype=torch.int64, embed_dtype=torch.float32, embed_dim=F, amp_mode=True,
		                 load_model=False, device="cpu")

	def tokenize(self, text, max_tokens=None, output_dict=False):
		texts = (text,) if isinstance(text, str) else tuple(text)
		rows = [[1] + [self.stoi[w] for w in t.split()] + [2] for t in texts]
		L = max(len(r) for r in rows)
		ids = torch.zeros(len(rows), L, dtype=torch.int64)
		att = torch.zeros(len(rows), L, dtype=torch.int64)
		for i, r in enumerate(rows):
			ids[i, :len(r)] = torch.tensor(r)
			att[i, :len(r)] = 1
		return {"input_ids": ids, "attention_mask": att} if output_dict else ids

	def detokenize(self, token_ids):
		def one(row):
			out = []
			for t in row.tolist():
				if t == 1:
					continue
				if t in (0, 2):
					break
				out.append(self.itos[t])
			return " ".join(out)
		return one(token_ids) if token_ids.ndim == 1 else [one(r) for r in token_ids]


def make_embedder():
	emb = LocalVocabEmbedder()
	tc = emb.create_target_config(NOUNS, with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True)
	emb.configure_target(tc, NOUNS)
	return emb


def write_cache(path, emb, N, M, full_targets, default_weights, seed):
	g = torch.Generator().manual_seed(seed)
	embeds = torch.nn.functional.normalize(torch.randn(N, F, generator=g), dim=-1)
	R = len(NOUNS)
	ids = torch.zeros(N, M, dtype=torch.int32)
	w = torch.zeros(N, M)
	for i in range(N):
		k = M if full_targets else int(torch.randint(1, M + 1, (1,), generator=g))
		ids[i, :k] = (torch.randperm(R, generator=g)[:k] + 1).int()
		ww = torch.rand(k, generator=g).sort(descending=True)[0] + 0.05
		w[i, :k] = ww / ww.sum()
	with ref_cache.EmbeddingCacheWriter(cache_path=path, embedder=emb, num_embed=N, shuffle=False, use_targets=True, full_targets=full_targets, target_nouns=NOUNS,
	                                    num_embed_targets=M, default_weights=default_weights, unit_weights=True, embedder_strict=True) as writer:
		for s in range(0, N, 7):
			writer.write(embeds=embeds[s:s + 7], embed_targets=ids[s:s + 7], embed_target_weights=None if default_weights else w[s:s + 7])


def read_batches(path, emb, batch_size, training, offset, data_kwargs):
	cache = ref_cache.EmbeddingCache(cache_path=path, embedder=emb, use_targets=True, strict_embedder=True)
	ds = cache.create_dataset(batch_size=batch_size, training=training)
	dc = ds.resolve_data_config(**data_kwargs)
	ds.configure_data(dc)
	out = []
	with ds.loaded():
		ds.epoch_index_offset = offset
		for i in range(len(ds)):
			out.append(tuple(None if t is None else t.clone() for t in ds[i]))
	import dataclasses
	return dict(batch_size=batch_size, training=training, offset=offset, data_kwargs=data_kwargs, data_config=dataclasses.asdict(dc), num_items=len(ds), batches=out)


def main():
	emb = make_embedder()
	specs = [("cache_single.bin", dict(N=37, M=1, full_targets=True, default_weights=True, seed=1)),
	         ("cache_multi.bin", dict(N=29, M=3, full_targets=False, default_weights=False, seed=2))]
	result = {}
	for name, kw in specs:
		path = os.path.join(HERE, name)
		if os.path.exists(path):
			os.remove(path)
		write_cache(path, emb, **kw)
		reads = []
		if kw["M"] == 1:
			reads.append(read_batches(path, emb, 8, False, 0, {}))
			reads.append(read_batches(path, emb, 8, True, 13, {}))
			reads.append(read_batches(path, emb, 5, True, 34, dict(use_weights=True)))
		else:
			reads.append(read_batches(path, emb, 6, False, 0, {}))
			reads.append(read_batches(path, emb, 6, True, 25, {}))
			reads.append(read_batches(path, emb, 6, True, 7, dict(multi_length=2)))
			reads.append(read_batches(path, emb, 4, True, 11, dict(multi_target=False)))
			reads.append(read_batches(path, emb, 4, False, 0, dict(multi_first=True, fixed_multi_length=True)))
			reads.append(read_batches(path, emb, 7, True, 3, dict(use_weights=False)))
		result[name] = dict(reads=reads, size=os.path.getsize(path))
		print(name, os.path.getsize(path), "bytes;", len(reads), "read configurations")
	result["tokens"], result["nouns"], result["embed_dim"] = TOKENS, NOUNS, F
	torch.save(result, os.path.join(HERE, "cache_batches.pt"))
	print("wrote cache_batches.pt", os.path.getsize(os.path.join(HERE, "cache_batches.pt")))


if __name__ == "__main__":
	main()
