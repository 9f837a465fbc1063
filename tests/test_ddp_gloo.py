"""World-size-2 rehearsal of the data-parallel gradient exchange on CPU (gloo): bucketed SUM all-reduce of a flat buffer,
parameter broadcast, statistic reduction -- the collectives bench.py / train_step issue over RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def _worker(rank, world, port, out):
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from novic_amd.train import DataParallel
	from novic_amd import ops
	dp = DataParallel(buckets=3, persistent_cus=240)
	assert dp.enabled and dp.world == world and dp.rank == rank
	g = torch.Generator().manual_seed(100 + rank)
	grad = torch.randn(10007, generator=g)
	mine = grad.clone()
	dp.begin_step()
	assert getattr(ops._tls, "cus", 0) == 0    # no collective in flight yet: the forward pass and the top of the backward pass keep every CU
	dp.reduce_range_early(grad, 7000, 9000)   # "layer 1" ready first, then "layer 0": reduced while the backward pass would still be running
	assert ops.current_cu_budget() == 240      # from the first early all-reduce on, this thread's persistent GEMM grids launch 16 workgroups short (per-call argument of the C ABI)
	dp.reduce_range_early(grad, 5000, 7000)
	dp.all_reduce_grads(grad)                  # the gaps [0, 5000) and [9000, end) + wait for everything
	assert getattr(ops._tls, "cus", 0) == 0    # ... and are back on the default once the exchange is complete
	dp.begin_step()
	again = mine.clone()
	dp.all_reduce_grads(again)                 # no early ranges: whole buffer in buckets
	assert torch.equal(again, grad)
	flat = torch.full((33,), float(rank))
	dp.broadcast_parameters(flat)
	stats = torch.tensor([[1.0, 2.0], [3.0 * (rank + 1), 4.0]])
	dp.all_reduce_stats(stats)
	out[rank] = (mine, grad, flat, stats)
	dist.barrier()
	dist.destroy_process_group()


def test_bucketed_gradient_all_reduce_world2():
	world, port = 2, _free_port()
	with mp.Manager() as mgr:
		out = mgr.dict()
		mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
		res = dict(out)
	total = res[0][0] + res[1][0]
	for r in range(world):
		torch.testing.assert_close(res[r][1], total)
		assert torch.all(res[r][2] == 0)
		assert res[r][3].tolist() == [[2.0, 4.0], [9.0, 8.0]]


# ---- world 4, an odd number of batches, stop + resume: every rank issues the same collectives ----

def _loader_worker(rank, world, port, out, resume_state):
	"""One rank of action_train's control flow without the GPU work: cache file -> rank-strided DeviceLoader (batch assembly stubbed: it is a HIP gather) -> GradAccum ->
	per optimizer step one DataParallel gradient exchange (early per-layer ranges + the rest) -> per chunk one statistics reduction; the loader state travels through a
	checkpoint dict as in training_loop.  Every dist.all_reduce / broadcast is counted."""
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
	dist.init_process_group("gloo", rank=rank, world_size=world)
	import sys
	here = os.path.dirname(os.path.abspath(__file__))
	sys.path.insert(0, here)
	from test_cache_reader import _embedder, GOLDEN
	from novic_amd import embedding_cache as EC, embedding_dataset
	from novic_amd.train import DataParallel
	calls = []
	real_all_reduce, real_broadcast = dist.all_reduce, dist.broadcast

	def counting_all_reduce(t, *a, **kw):
		calls.append(("all_reduce", t.numel()))
		return real_all_reduce(t, *a, **kw)

	def counting_broadcast(t, *a, **kw):
		calls.append(("broadcast", t.numel()))
		return real_broadcast(t, *a, **kw)
	dist.all_reduce, dist.broadcast = counting_all_reduce, counting_broadcast
	dp = DataParallel(buckets=2)
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, "cache_single.bin"), _embedder("cpu"), strict_embedder=True)
	ds = cache.create_dataset(batch_size=1, training=True)  # 37 batches: 37 % 4 = 1 left out per epoch, 9 per rank, accum 2 drops one more -> 4 optimizer steps per epoch
	ds.configure_data(ds.resolve_data_config())
	loader = EC.DeviceLoader(ds, torch.device("cpu"), seed=11, rank=rank, world=world)
	loader.assemble = lambda index, slot=None: index  # the batch's position in the shuffled order stands for the batch
	ga = embedding_dataset.GradAccum(loader, loader.loader_info, accum_size=2, drop_last=True)
	flat = torch.zeros(5000)
	dp.broadcast_parameters(flat)
	steps_total, chunk_steps = 0, 2
	seen, lrs = [], []
	state = dict(step=0, epoch=0)
	if resume_state is not None:  # what action_train restores from the .train file: loader RNG, counters (the same file on every rank)
		loader.load_state_dict(resume_state["loader"])
		state = dict(resume_state["loop"])
	stop_at = None if resume_state is not None else 6  # first run: stop after 6 optimizer steps = in the middle of the second epoch
	saved = None
	while state["epoch"] < 4 and saved is None:
		epoch_batches = []
		n_in_step = 0
		for index in ga.loader():
			epoch_batches.append(index)
			_, step = ga.loss_scale(1)
			n_in_step += 1
			if step:
				grad = torch.full((5000,), float(rank + 1))
				dp.begin_step()
				dp.reduce_range_early(grad, 3000, 4000)  # two "layers" reduced inside the backward pass
				dp.reduce_range_early(grad, 2000, 3000)
				dp.all_reduce_grads(grad)
				assert float(grad[0]) == sum(range(1, world + 1)) and float(grad[2500]) == sum(range(1, world + 1))
				state["step"] += 1
				n_in_step = 0
				if state["step"] % chunk_steps == 0:  # chunk boundary: statistics reduction, checkpoint opportunity
					stats = torch.ones(4, 2)
					dp.all_reduce_stats(stats)
					assert float(stats[0, 0]) == world
				if stop_at is not None and state["step"] == stop_at:
					saved = dict(loader=loader.state_dict(), loop=dict(state, epoch=state["epoch"] + 1))  # the interrupted epoch is not replayed (reference: GradAccum state is not saved)
					break
		seen.append(epoch_batches)
		state["epoch"] += 1
	out[rank] = dict(calls=list(calls), seen=seen, saved=saved, steps=state["step"])
	dist.barrier()
	dist.destroy_process_group()
	dist.all_reduce, dist.broadcast = real_all_reduce, real_broadcast


def test_world4_odd_batches_stop_and_resume_issue_equal_collectives():
	world = 4
	runs = []
	saved = None
	for phase in range(2):
		port = _free_port()
		with mp.Manager() as mgr:
			out = mgr.dict()
			mp.spawn(_loader_worker, args=(world, port, out, saved), nprocs=world, join=True)
			res = {r: dict(v) for r, v in dict(out).items()}
		runs.append(res)
		# every rank issued the SAME sequence of collectives (kind and size): nothing can pair up mismatched or hang
		assert all(res[r]["calls"] == res[0]["calls"] for r in range(world)), phase
		assert len(res[0]["calls"]) > 0 and res[0]["steps"] == res[1]["steps"] == res[2]["steps"] == res[3]["steps"]
		# within an epoch the ranks stride ONE shuffled order: disjoint batches, together a prefix of it; every rank gets the same count
		for e in range(len(res[0]["seen"])):
			per_rank = [res[r]["seen"][e] for r in range(world)]
			assert len({len(p) for p in per_rank}) == 1
			flat = [i for p in per_rank for i in p]
			assert len(flat) == len(set(flat)) and all(0 <= i < 37 for i in flat)
		if phase == 0:
			assert all(res[r]["saved"] is not None for r in range(world))
			assert all(res[r]["saved"]["loader"] == res[0]["saved"]["loader"] for r in range(world))  # one checkpoint serves every rank
			assert res[0]["steps"] == 6 and len(res[0]["seen"]) == 2 and len(res[0]["seen"][0]) == 8 and len(res[0]["seen"][1]) == 4  # stopped in the middle of epoch 2
			saved = res[0]["saved"]
	# the resumed run continues with fresh epochs drawn from the restored generator: identical on every rank, and different from a restart from scratch
	first, second = runs
	assert len(second[0]["seen"]) == 2 and all(len(e) == 8 for e in second[0]["seen"]) and second[0]["steps"] == 6 + 2 * 4
	assert second[0]["seen"][0] not in (first[0]["seen"][0], first[0]["seen"][1] + second[0]["seen"][0][4:])


# ---- configs[4] in small: the multiset step (M weighted targets per embedding) through the data-parallel control flow at world 2 ----

def _multiset_worker(rank, world, port, out):
	"""Rank-strided loader over the reference-written multi-target cache (cache_multi.bin: M targets + weights per embedding), batches assembled by the HOST reader (the
	device loader's gather is a HIP kernel), GradAccum, and per optimizer step one gradient exchange of a flat buffer laid out like the F = 1024 decoder's (prefix MLP
	2048 x 1024 first, then the tied embedding, six layers of 1 179 648, the norms): the "gradient" is a deterministic function of the batches a rank consumed, so the
	reduced result can be checked against a single process that consumes every batch."""
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
	dist.init_process_group("gloo", rank=rank, world_size=world)
	import sys
	sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
	from test_cache_reader import _embedder, GOLDEN
	from novic_amd import embedding_cache as EC, embedding_dataset, ops
	from novic_amd.train import DataParallel
	dp = DataParallel(buckets=2, persistent_cus=232)
	cache = EC.EmbeddingCache(os.path.join(GOLDEN, "cache_multi.bin"), _embedder("cpu"), strict_embedder=True)
	ds = cache.create_dataset(batch_size=4, training=True)
	dc = ds.resolve_data_config()
	ds.configure_data(dc)
	assert dc.multi_target and dc.use_weights
	host_batches = {}
	loader = EC.DeviceLoader(ds, torch.device("cpu"), seed=5, rank=rank, world=world)
	loader.assemble = lambda index, slot=None: index
	ga = embedding_dataset.GradAccum(loader, loader.loader_info, accum_size=2, drop_last=True)
	layout = [("prefix", 2048 * 1024), ("tok", 6912 * 512), ("pos", 15 * 512)] + [(f"layer{i}", 1179648) for i in range(6)] + [("norms", 13 * 512)]
	total = sum(n for _, n in layout)
	offs, pos = {}, 0
	for name, n in layout:
		offs[name] = (pos, pos + n)
		pos += n
	consumed, reduced = [], []
	acc = torch.zeros(total)
	for index in ga.loader():
		consumed.append(int(index))
		acc += float(index + 1)  # this micro-batch's "gradient": every element gets (position of the batch in the epoch's order + 1)
		_, step = ga.loss_scale(1)
		if step:
			dp.begin_step()
			for i in reversed(range(6)):  # the layers' ranges leave first, the last layer first, as in the backward pass
				dp.reduce_range_early(acc, *offs[f"layer{i}"])
			assert ops.current_cu_budget() == 232
			dp.all_reduce_grads(acc)
			reduced.append(float(acc[0]))
			assert float(acc[offs["layer3"][0]]) == reduced[-1] and float(acc[-1]) == reduced[-1]
			acc = torch.zeros(total)
	out[rank] = dict(consumed=consumed, reduced=reduced, n=len(ds))
	dist.barrier()
	dist.destroy_process_group()


def test_multiset_step_world2():
	world, port = 2, _free_port()
	with mp.Manager() as mgr:
		out = mgr.dict()
		mp.spawn(_multiset_worker, args=(world, port, out), nprocs=world, join=True)
		res = {r: dict(v) for r, v in dict(out).items()}
	a, b = res[0]["consumed"], res[1]["consumed"]
	assert len(a) == len(b) > 0 and not set(a) & set(b)  # equal counts on both ranks, disjoint batches of one shuffled order
	assert res[0]["reduced"] == res[1]["reduced"] and len(res[0]["reduced"]) == len(a) // 2
	# every optimizer step's exchanged value is the sum over both ranks' two micro-batches of that step: what one process consuming all four would have accumulated
	for k, v in enumerate(res[0]["reduced"]):
		assert v == sum(i + 1 for i in a[2 * k:2 * k + 2] + b[2 * k:2 * k + 2])


# ---- world 8, the size configs[2] names: three optimizer steps with a stop + resume after the second, parameters bit-equal on every rank ----

def _world8_worker(rank, world, port, out, resume, steps):
	"""train_step's data-parallel arithmetic without the HIP kernels: every rank holds the same flat parameters (broadcast from rank 0, or restored from ONE checkpoint),
	computes a rank-specific gradient from them, exchanges it as train_step does (two early per-layer ranges out of order + the bucketed rest, 2 M elements so that the
	gaps really split into buckets), clips by the post-reduce global norm and updates.  Counts every collective."""
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
	dist.init_process_group("gloo", rank=rank, world_size=world)
	from novic_amd.train import DataParallel
	calls = []
	real_all_reduce, real_broadcast = dist.all_reduce, dist.broadcast

	def counting_all_reduce(t, *a, **kw):
		calls.append(("all_reduce", t.numel()))
		return real_all_reduce(t, *a, **kw)

	def counting_broadcast(t, *a, **kw):
		calls.append(("broadcast", t.numel()))
		return real_broadcast(t, *a, **kw)
	dist.all_reduce, dist.broadcast = counting_all_reduce, counting_broadcast
	dp = DataParallel(buckets=2, persistent_cus=240)
	N = 3 << 20
	if resume is None:
		flat = torch.randn(N, generator=torch.Generator().manual_seed(7 + rank))  # (every rank starts with its OWN values: the broadcast must make them rank 0's)
		first = 0
	else:
		flat = resume["flat"].clone()
		first = resume["step"]
	dp.broadcast_parameters(flat)
	for step in range(first, steps):
		g = torch.Generator().manual_seed(1000 * step + rank)
		grad = (torch.randn(N, generator=g) + 0.01 * flat) / world  # rank-specific, depends on the parameters; pre-scaled by 1 / world as train_step's loss is
		dp.begin_step()
		dp.reduce_range_early(grad, 2 << 20, (2 << 20) + 300000)
		dp.reduce_range_early(grad, 1 << 20, (1 << 20) + 500000)
		dp.all_reduce_grads(grad)
		norm = grad.double().norm()  # post-reduce: the same number on every rank, no extra collective
		grad.mul_(float(min(1.0, 1.0 / (float(norm) + 1e-6))))
		flat.add_(grad, alpha=-0.05)
		stats = torch.tensor([[1.0, float(rank)]])
		dp.all_reduce_stats(stats)
		assert stats.tolist() == [[float(world), float(sum(range(world)))]]
	out[rank] = dict(calls=list(calls), flat=flat.clone(), step=steps)
	dist.barrier()
	dist.destroy_process_group()
	dist.all_reduce, dist.broadcast = real_all_reduce, real_broadcast


def _spawn8(resume, steps):
	world, port = 8, _free_port()
	with mp.Manager() as mgr:
		out = mgr.dict()
		mp.spawn(_world8_worker, args=(world, port, out, resume, steps), nprocs=world, join=True)
		return {r: dict(v) for r, v in dict(out).items()}


def test_world8_three_steps_with_a_resume_keep_parameters_bit_equal():
	straight = _spawn8(None, 3)
	for r in range(8):
		assert straight[r]["calls"] == straight[0]["calls"], r                    # the same collectives in the same order on every rank: nothing can pair up mismatched
		assert torch.equal(straight[r]["flat"], straight[0]["flat"]), r            # ... and bit-equal parameters after three steps
	kinds = [k for k, _ in straight[0]["calls"]]
	assert kinds.count("broadcast") == 1 and kinds.count("all_reduce") == 3 * (2 + (2 + 1 + 1) + 1)  # per step: two early ranges; three gaps -- [0, 1 Mi) in two buckets, the two shorter ones whole; one statistics reduction
	stopped = _spawn8(None, 2)
	ckpt = dict(flat=stopped[3]["flat"], step=2)                                    # any rank's copy serves: they are bit-equal
	assert all(torch.equal(stopped[r]["flat"], stopped[0]["flat"]) for r in range(8))
	resumed = _spawn8(ckpt, 3)
	for r in range(8):
		assert torch.equal(resumed[r]["flat"], straight[0]["flat"]), r             # stop after two steps + resume = the uninterrupted run, on every rank
