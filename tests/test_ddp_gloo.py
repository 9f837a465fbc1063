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
	dp = DataParallel(buckets=3)
	assert dp.enabled and dp.world == world and dp.rank == rank
	g = torch.Generator().manual_seed(100 + rank)
	grad = torch.randn(10007, generator=g)
	mine = grad.clone()
	dp.begin_step()
	dp.reduce_range_early(grad, 7000, 9000)   # "layer 1" ready first, then "layer 0": reduced while the backward pass would still be running
	dp.reduce_range_early(grad, 5000, 7000)
	dp.all_reduce_grads(grad)                  # the gaps [0, 5000) and [9000, end) + wait for everything
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
