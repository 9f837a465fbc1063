"""Embedding-cache reader against cache files written by the reference's own writer and batches read back by the reference's own
Dataset.__getitem__ (tests/golden/make_golden_cache.py): host path on CPU (bit-exact), device gather path on the GPU (bit-exact)."""
import dataclasses
import os
import threading

import pytest
import torch

from conftest import GOLDEN, load_golden

GOLD = load_golden("cache_batches.pt")
FILES = ("cache_single.bin", "cache_multi.bin")


def _embedder(device="cpu"):
	from novic_amd import embedders
	from novic_amd.embedding_decoder import PrefixedIterDecoder
	emb = embedders.LocalVocabEmbedder(tokens=GOLD["tokens"], embed_dim=GOLD["embed_dim"], device=device)
	tc = emb.create_target_config(GOLD["nouns"], **PrefixedIterDecoder.get_target_config_kwargs(with_start_token=True, with_end_token=False, compact_ids=False,
	                                                                                             fixed_token_length=False, auto_fixed_token_length=True, use_masks=True))
	emb.configure_target(tc, GOLD["nouns"])
	return emb


def _same(a, b):
	if a is None or b is None:
		return a is None and b is None
	return a.shape == b.shape and a.dtype == b.dtype and torch.equal(a, b)


@pytest.mark.parametrize("fname", FILES)
def test_host_reader_matches_reference_batches(fname):
	from novic_amd import embedding_cache as EC
	emb = _embedder("cpu")
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, fname), emb, strict_embedder=True)  # hashes of the twin embedder configuration must match the file's
	assert cache.cache_size == GOLD[fname]["size"] and cache.target_nouns == ("",) + tuple(GOLD["nouns"])
	for read in GOLD[fname]["reads"]:
		ds = cache.create_dataset(batch_size=read["batch_size"], training=read["training"])
		dc = ds.resolve_data_config(**read["data_kwargs"])
		assert dataclasses.asdict(dc) == read["data_config"]
		ds.configure_data(dc)
		assert len(ds) == read["num_items"]
		with ds.loaded():
			ds.epoch_index_offset = read["offset"]
			for i, ref in enumerate(read["batches"]):
				got = ds[i]
				for g, r in zip(got, ref):
					assert _same(g, r), (fname, read["data_kwargs"], i)
			with pytest.raises(IndexError):
				ds[len(ds)]


def test_reader_rejects_damaged_or_mismatched_files(tmp_path):
	from novic_amd import embedders, embedding_cache as EC
	emb = _embedder("cpu")
	raw = open(os.path.join(GOLDEN, "cache_single.bin"), "rb").read()
	bad = tmp_path / "bad_magic.bin"
	bad.write_bytes(b"\x00" * 32 + raw[32:])
	with pytest.raises(ValueError, match="magic"):
		EC.EmbeddingCache(str(bad), emb)
	short = tmp_path / "short.bin"
	short.write_bytes(raw[:-4])
	with pytest.raises(ValueError, match="size"):
		EC.EmbeddingCache(str(short), emb)
	other = embedders.LocalVocabEmbedder(tokens=GOLD["tokens"] + ["extra"], embed_dim=GOLD["embed_dim"], device="cpu")
	with pytest.raises(ValueError, match="hash"):
		EC.EmbeddingCache(os.path.join(GOLDEN, "cache_single.bin"), other, strict_embedder=True)
	wrong_dim = embedders.LocalVocabEmbedder(tokens=GOLD["tokens"], embed_dim=32, device="cpu")
	with pytest.raises(ValueError, match="dimension"):
		EC.EmbeddingCache(os.path.join(GOLDEN, "cache_single.bin"), wrong_dim, strict_embedder=False)
	with pytest.raises(RuntimeError):
		EC.EmbeddingCache(os.path.join(GOLDEN, "cache_single.bin"), emb).get_samples(0, 4)  # not entered


def test_data_parallel_loader_gives_every_rank_the_same_number_of_batches():
	"""rank-strided loaders of different lengths would pair up mismatched all-reduces.  Host logic only (no batch is assembled)."""
	from novic_amd import embedding_cache as EC, embedding_dataset
	emb = _embedder("cpu")
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, "cache_single.bin"), emb, strict_embedder=True)
	ds = cache.create_dataset(batch_size=4, training=True)  # 37 embeddings -> 9 training batches: odd
	ds.configure_data(ds.resolve_data_config())
	assert ds.num_items == 9
	cpu = torch.device("cpu")
	for world in (2, 4):
		loaders = [EC.DeviceLoader(ds, cpu, seed=3, rank=r, world=world) for r in range(world)]
		assert {len(ld) for ld in loaders} == {9 // world}
		for ld in loaders:
			li = ld.loader_info
			assert li.epoch_batches == len(ld) == li.complete_batches and li.epoch_samples == len(ld) * 4 and not li.incomplete_batch
			ga = embedding_dataset.GradAccum(ld, li, accum_size=2, drop_last=True)  # what action_train builds per rank
			assert ga.loader_batches == (9 // world) // 2 * 2
	with pytest.raises(ValueError, match="seed"):
		EC.DeviceLoader(ds, cpu, rank=0, world=2)  # OS entropy per rank would shuffle the ranks differently
	with pytest.raises(ValueError):
		EC.DeviceLoader(ds, cpu, seed=1, rank=2, world=2)
	one = EC.DeviceLoader(ds, cpu, seed=3)
	assert len(one) == 9 and one.loader_info == ds.loader_info
	# evaluation keeps every batch (no collective depends on the count)
	ev = cache.create_dataset(batch_size=4, training=False)
	ev.configure_data(ev.resolve_data_config())
	assert sum(len(EC.DeviceLoader(ev, cpu, seed=0, rank=r, world=2)) for r in range(2)) == ev.num_items == 10
	# the shuffle generator is checkpointable
	a, b = EC.DeviceLoader(ds, cpu, seed=9), EC.DeviceLoader(ds, cpu, seed=123)
	b.load_state_dict(a.state_dict())
	assert a.rng.random() == b.rng.random()


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
def test_device_loader_matches_reference_batches(fname):
	from novic_amd import embedding_cache as EC
	emb = _embedder("cuda")
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, fname), emb, strict_embedder=False)  # device_type differs from the writer's ('cpu'): dimension/dtype checks only
	for read in GOLD[fname]["reads"]:
		ds = cache.create_dataset(batch_size=read["batch_size"], training=read["training"])
		ds.configure_data(ds.resolve_data_config(**read["data_kwargs"]))
		loader = EC.DeviceLoader(ds, torch.device("cuda"), seed=0)
		ds.epoch_index_offset = read["offset"]
		for i, ref in enumerate(read["batches"]):
			got = loader.assemble(i)
			for g, r in zip(got, ref):
				assert _same(None if g is None else g.cpu(), r), (fname, read["data_kwargs"], i)
	# epoch iteration: every batch of the (shuffled, rotated) epoch exactly once; two ranks split it without overlap
	ds = cache.create_dataset(batch_size=4, training=True)
	ds.configure_data(ds.resolve_data_config())
	full = EC.DeviceLoader(ds, torch.device("cuda"), seed=5)
	seen = [b[0].cpu() for b in full]
	assert len(seen) == len(full) == ds.num_items
	r0 = [b[0].cpu() for b in EC.DeviceLoader(ds, torch.device("cuda"), seed=5, rank=0, world=2)]
	r1 = [b[0].cpu() for b in EC.DeviceLoader(ds, torch.device("cuda"), seed=5, rank=1, world=2)]
	assert len(r0) == len(r1) == len(seen) // 2  # 9 batches: the odd one at the end of the shuffled order is left out, every rank runs the same number of steps
	merged = [None] * (2 * len(r0))
	merged[0::2], merged[1::2] = r0, r1
	assert all(torch.equal(a, b) for a, b in zip(merged, seen))


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
@pytest.mark.parametrize("training", [False, True])
def test_streaming_loader_yields_the_resident_loaders_batches(fname, training):
	"""A cache beyond the HBM budget (forced: budget 1 byte) streams its embedding rows through pinned staging buffers and device slabs; every batch
	of an epoch -- shuffled order, rotation, wrap-around, two ranks -- equals the resident loader's bit for bit."""
	from novic_amd import embedding_cache as EC
	emb = _embedder("cuda")
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, fname), emb, strict_embedder=False)
	for batch_size, depth in ((4, 2), (5, 3), (7, 4)):
		ds = cache.create_dataset(batch_size=batch_size, training=training)
		ds.configure_data(ds.resolve_data_config())
		for rank, world in ((0, 1), (1, 2)):
			resident = EC.DeviceLoader(ds, torch.device("cuda"), seed=11, rank=rank, world=world)
			streamed = EC.DeviceLoader(ds, torch.device("cuda"), seed=11, rank=rank, world=world, hbm_budget_bytes=1, stream_depth=depth)
			assert streamed.streaming and not resident.streaming and streamed.embeds is None
			for epoch in range(2):  # the second epoch reuses every staging slot
				a = [tuple(None if t is None else t.cpu() for t in b) for b in resident]
				b = [tuple(None if t is None else t.cpu() for t in b) for b in streamed]
				assert len(a) == len(b) == len(resident) and len(a) > 0
				for x, y in zip(a, b):
					assert all(_same(p, q) for p, q in zip(x, y)), (fname, training, batch_size, rank, epoch)


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
@pytest.mark.parametrize("training", [False, True])
def test_grouped_loader_yields_the_same_batches(fname, training):
	"""DeviceLoader(group = G): G loader batches -- an optimizer step's micro-batches -- assembled by ONE gather launch into one set of buffers (novic_cache_gather_group) and
	handed out as views.  Every batch of an epoch (shuffled order, rotation, wrap-around of the row window, a remainder shorter than a group, the short last batch, two ranks,
	every data configuration of the golden reads) equals the ungrouped loader's bit for bit; the slices know their group (train_step then skips its concatenation)."""
	from novic_amd import embedding_cache as EC
	emb = _embedder("cuda")
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, fname), emb, strict_embedder=False)
	configs = [{}] + [read["data_kwargs"] for read in GOLD[fname]["reads"]]
	grouped_any = False
	for kw in configs:
		for batch_size, G in ((4, 2), (3, 4), (5, 3)):
			ds = cache.create_dataset(batch_size=batch_size, training=training)
			ds.configure_data(ds.resolve_data_config(**kw))
			for rank, world in ((0, 1), (1, 2)):
				plain = EC.DeviceLoader(ds, torch.device("cuda"), seed=11, rank=rank, world=world)
				a = [tuple(None if t is None else t.cpu() for t in b) for b in plain]
				# resident, and STREAMING (budget forced to 1 byte): there a staging thread fills step slabs two groups ahead and each group is one gather out of its slab
				for grouped in (EC.DeviceLoader(ds, torch.device("cuda"), seed=11, rank=rank, world=world, group=G),
				                EC.DeviceLoader(ds, torch.device("cuda"), seed=11, rank=rank, world=world, group=G, hbm_budget_bytes=1, stream_depth=3)):
					for epoch in range(2 if grouped.streaming else 1):  # (the second epoch reuses every step slab; the plain loader's rng advances alike)
						if epoch:
							a = [tuple(None if t is None else t.cpu() for t in b) for b in plain]
						raw = list(grouped)
						b = [tuple(None if t is None else t.cpu() for t in bb) for bb in raw]
						assert len(a) == len(b) == len(plain) and len(a) > 0
						for x, y in zip(a, b):
							assert all(_same(p, q) for p, q in zip(x, y)), (fname, training, kw, batch_size, G, rank, grouped.streaming, epoch)
						slices = [bb for bb in raw if isinstance(bb, EC.GroupSlice)]
						grouped_any |= bool(slices)
						for s in slices:
							assert s.size == G and 0 <= s.pos < G and s[0].data_ptr() == s.full[0].data_ptr() + s.pos * s[0].numel() * 4
					if grouped.streaming:  # leaving an epoch early must stop the staging thread (generator close -> finally)
						it = iter(grouped)
						next(it)
						it.close()
						assert not [t for t in threading.enumerate() if t.name == "novic-loader-stage"]
	assert grouped_any


@pytest.mark.gpu
@pytest.mark.parametrize("fname", FILES)
def test_grouped_streaming_loader_across_epochs_without_host_synchronisation(fname):
	"""The step slabs of the grouped streaming loader outlive an epoch, and so must the events that guard them: training_loop synchronises per chunk, not per epoch, and
	the host runs ahead of the device, so the gathers of epoch N's last groups can still be queued when epoch N + 1 stages its first groups into the same slabs.  Here the
	compute stream is held back by a long spin kernel in front of each epoch's last gathers, two epochs are enqueued back to back with no host synchronisation (device-side clones keep each batch), and every
	batch of both epochs must equal the plain resident loader's.  With per-iteration events the copies of epoch 2 overtook the gathers of epoch 1 (advisor, round 4)."""
	from novic_amd import embedding_cache as EC
	emb = _embedder("cuda")
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, fname), emb, strict_embedder=False)
	ds = cache.create_dataset(batch_size=2, training=True)
	ds.configure_data(ds.resolve_data_config())
	plain = EC.DeviceLoader(ds, torch.device("cuda"), seed=5)
	grouped = EC.DeviceLoader(ds, torch.device("cuda"), seed=5, group=2, hbm_budget_bytes=1, stream_depth=3)
	assert grouped.streaming and len(grouped) // 2 >= 4  # more groups than step slabs: every slab is reused inside an epoch and again at the head of the next
	want = [[tuple(None if t is None else t.cpu() for t in b) for b in plain] for _ in range(3)]
	torch.cuda.synchronize()
	got, G, S = [], 2, 3
	nfull = len(grouped) // G * G
	for epoch in range(2):
		batches = []
		for k, b in enumerate(grouped):
			if k == nfull - (S + 1) * G:  # (a group's gather is enqueued before its first batch is yielded) the gathers of the epoch's last S groups are enqueued behind ~0.2 s of spinning: still pending when the next epoch stages its first groups
				torch.cuda._sleep(int(4e8))
			batches.append(tuple(None if t is None else t.clone() for t in b))
		got.append(batches)
	it = iter(grouped)  # ... and an early close followed by a new iteration, still behind the spin kernel's successors
	first = tuple(None if t is None else t.clone() for t in next(it))
	it.close()
	torch.cuda.synchronize()
	for epoch in range(2):
		assert len(got[epoch]) == len(want[epoch]) > 0
		for k, (x, y) in enumerate(zip(want[epoch], got[epoch])):
			assert all(_same(p, None if q is None else q.cpu()) for p, q in zip(x, y)), (fname, epoch, k)
	assert all(_same(p, None if q is None else q.cpu()) for p, q in zip(want[2][0], first))
