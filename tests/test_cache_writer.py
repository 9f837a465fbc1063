"""Embedding-cache WRITER (novic_amd.embedding_cache.EmbeddingCacheWriter; reference embedding_cache.py:161-459) against the files the reference's
own writer produced: tests/golden/cache_single.bin / cache_multi.bin were written by the reference from seeded inputs
(tests/golden/make_golden_cache.py::write_cache); the same inputs through this writer must give the same bytes.  Plus the writer's own rules
(magic bytes only on a complete file, shuffling, validation) and, on the GPU, PhotoCacheWriter through the tokenizer and the native text tower."""
import os

import pytest
import torch

from conftest import GOLDEN, load_golden
from test_cache_reader import _embedder

GOLD = load_golden("cache_batches.pt")
SPECS = (("cache_single.bin", dict(N=37, M=1, full_targets=True, default_weights=True, seed=1)),
         ("cache_multi.bin", dict(N=29, M=3, full_targets=False, default_weights=False, seed=2)))


def _inputs(N, M, full_targets, seed, F, R):
	"""The seeded inputs of tests/golden/make_golden_cache.py::write_cache (same generator call sequence)."""
	g = torch.Generator().manual_seed(seed)
	embeds = torch.nn.functional.normalize(torch.randn(N, F, generator=g), dim=-1)
	ids = torch.zeros(N, M, dtype=torch.int32)
	w = torch.zeros(N, M)
	for i in range(N):
		k = M if full_targets else int(torch.randint(1, M + 1, (1,), generator=g))
		ids[i, :k] = (torch.randperm(R, generator=g)[:k] + 1).int()
		ww = torch.rand(k, generator=g).sort(descending=True)[0] + 0.05
		w[i, :k] = ww / ww.sum()
	return embeds, ids, w


@pytest.mark.parametrize("fname,kw", SPECS)
def test_writer_reproduces_the_reference_writers_file(tmp_path, fname, kw):
	from novic_amd import embedding_cache as EC
	emb = _embedder("cpu")
	nouns = tuple(GOLD["nouns"])
	embeds, ids, w = _inputs(kw["N"], kw["M"], kw["full_targets"], kw["seed"], GOLD["embed_dim"], len(nouns))
	path = str(tmp_path / fname)
	with EC.EmbeddingCacheWriter(cache_path=path, embedder=emb, num_embed=kw["N"], shuffle=False, use_targets=True, full_targets=kw["full_targets"], target_nouns=nouns,
	                             num_embed_targets=kw["M"], default_weights=kw["default_weights"], unit_weights=True, embedder_strict=True) as writer:
		assert writer.tensorize_embed_targets(["dog", ("sea", "cat")][:1 if kw["M"] == 1 else 2]).shape[1] == kw["M"]
		for s in range(0, kw["N"], 7):
			writer.write(embeds=embeds[s:s + 7], embed_targets=ids[s:s + 7], embed_target_weights=None if kw["default_weights"] else w[s:s + 7])
	assert open(path, "rb").read() == open(os.path.join(GOLDEN, fname), "rb").read()


def test_incomplete_or_invalid_writes_leave_no_file(tmp_path):
	from novic_amd import embedding_cache as EC
	emb = _embedder("cpu")
	nouns = tuple(GOLD["nouns"])
	embeds, ids, _ = _inputs(10, 1, True, 5, GOLD["embed_dim"], len(nouns))
	mk = lambda p, **kw: EC.EmbeddingCacheWriter(cache_path=str(p), embedder=emb, num_embed=10, shuffle=False, target_nouns=nouns, default_weights=True, **kw)
	with pytest.raises(RuntimeError):  # fewer embeddings than announced
		with mk(tmp_path / "a.bin") as wr:
			wr.write(embeds=embeds[:6], embed_targets=ids[:6])
	assert not os.path.exists(tmp_path / "a.bin")
	with pytest.raises(ValueError):  # not unit vectors
		with mk(tmp_path / "b.bin") as wr:
			wr.write(embeds=embeds * 1.01, embed_targets=ids)
	assert not os.path.exists(tmp_path / "b.bin")
	with pytest.raises(ValueError):  # a zero target id although full targets were promised
		with mk(tmp_path / "c.bin") as wr:
			wr.write(embeds=embeds, embed_targets=torch.zeros_like(ids))
	assert not os.path.exists(tmp_path / "c.bin")
	with pytest.raises(ValueError):  # weights given although default weights were promised
		with mk(tmp_path / "d.bin") as wr:
			wr.write(embeds=embeds, embed_targets=ids, embed_target_weights=torch.ones(10, 1))
	with pytest.raises(ValueError):
		EC.EmbeddingCacheWriter(cache_path=str(tmp_path / "e.bin"), embedder=emb, num_embed=3, target_nouns=("dog", "dog"))  # duplicate nouns


def test_shuffled_and_targetless_caches_read_back(tmp_path):
	from novic_amd import embedding_cache as EC
	emb = _embedder("cpu")
	nouns = tuple(GOLD["nouns"])
	N = 23
	embeds, ids, _ = _inputs(N, 1, True, 9, GOLD["embed_dim"], len(nouns))
	path = str(tmp_path / "shuffled.bin")
	torch.manual_seed(4)
	with EC.EmbeddingCacheWriter(cache_path=path, embedder=emb, num_embed=N, shuffle=True, target_nouns=nouns, default_weights=True) as wr:
		perm = wr.shuffle_perm.clone().long()
		for s in range(0, N, 5):
			wr.write(embeds=embeds[s:s + 5], embed_targets=ids[s:s + 5])
	assert sorted(perm.tolist()) == list(range(N)) and perm.tolist() != list(range(N))
	with EC.EmbeddingCache(path, emb, strict_embedder=True) as cache:
		got_e, got_t = cache.get_samples(0, N)[:2]
		assert torch.equal(got_e[perm], embeds) and torch.equal(got_t[perm].view(-1), ids.view(-1).to(got_t.dtype))  # sample i landed at row perm[i]
	rpath = str(tmp_path / "random.bin")
	torch.manual_seed(5)
	EC.RandomCacheWriter(rpath, emb, num_embed=50, batch_size=16).generate()
	with EC.EmbeddingCache(rpath, emb, use_targets=False, strict_embedder=False) as cache:
		e = cache.get_samples(0, 50)[0]
		assert e.shape == (50, GOLD["embed_dim"]) and torch.allclose(e.norm(dim=1), torch.ones(50), atol=1e-6)


@pytest.mark.gpu
def test_photo_cache_writer_through_tokenizer_and_native_text_tower(tmp_path):
	"""'a photo of a NOUN' prompts (reference embedding_cache_writers.py:50-104) over the local Hugging Face fixture directory: every noun's row of the
	cache is the text tower's embedding of its prompt, its target id is the noun's own, and the token table is tokenize_target's."""
	from novic_amd import embedders, embedding_cache as EC
	from novic_amd.embedding_decoder import PrefixedIterDecoder
	emb = embedders.Embedder.create("transformers:" + os.path.join(GOLDEN, "hf_clip_tiny"), device="cuda", inference_batch_size=3)
	nouns = ("cat", "dog", "bird house", "starling", "photo", "the ant", "house")
	tc = emb.create_target_config(nouns, **PrefixedIterDecoder.get_target_config_kwargs(with_start_token=True, with_end_token=False, compact_ids=False, fixed_token_length=False,
	                                                                                    auto_fixed_token_length=True, use_masks=True))
	emb.configure_target(tc, nouns)
	path = str(tmp_path / "photo.bin")
	all_embeds, tok, mask = EC.PhotoCacheWriter(path, emb, nouns, debug=True, shuffle=False).generate()
	with emb.inference_model(), emb.inference_mode():
		want = emb.inference_text(tuple(f"a photo of a {n}" for n in nouns)).cpu()
	assert float((all_embeds - want).norm(dim=1).max()) <= 1e-3  # batches of 3 vs one batch of 7: the same kernels on other row counts
	ids, m = emb.tokenize_target(nouns)
	assert torch.equal(tok, ids) and torch.equal(mask, m)
	with EC.EmbeddingCache(path, emb, strict_embedder=True) as cache:
		e, t = cache.get_samples(0, len(nouns))[:2]
		assert torch.equal(e, all_embeds) and t.view(-1).tolist() == list(range(1, len(nouns) + 1))
		assert cache.target_nouns == ("",) + nouns
