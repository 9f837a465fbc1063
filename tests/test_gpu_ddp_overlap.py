"""Two ranks on the one GPU of the test box (gloo carries CUDA tensors through the host): train_step with the per-layer early gradient reduction must
give every rank the parameters a single process gets from the same micro-batches (the data-parallel contract of SURVEY 8e), and both ranks the
same bits."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		return s.getsockname()[1]


def _setup(seed):
	import sys
	sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
	from helpers import make_decoder, synth_batch, to_dev
	from oracle import decoder_oracle as O
	from novic_amd import train as T
	spec = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=32, num_layers=2, num_heads=4)
	model, _ = make_decoder(spec, seed=seed, dropout=0.0, device="cuda")
	model.train()
	opt = T.FusedAdamW(model, lr=1e-2, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	mbs = [to_dev(*synth_batch(spec, 16, seed=50 + i, max_len=5)) for i in range(4)]
	C = max(mb[1].shape[1] for mb in mbs)  # same width so that the micro-batches merge
	mbs = [(e, torch.nn.functional.pad(t, (0, C - t.shape[1])), torch.nn.functional.pad(m, (0, C - m.shape[1]), value=True), w) for e, t, m, w in mbs]
	return T, model, opt, mbs


def _worker(rank, world, port, out):
	os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
	torch.cuda.set_device(0)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	T, model, opt, mbs = _setup(seed=7)
	dp = T.DataParallel()
	assert dp.enabled
	calls = []
	orig = dp.reduce_range_early
	dp.reduce_range_early = lambda g, s, e: (calls.append((s, e)), orig(g, s, e))[1]
	for _ in range(2):
		T.train_step(model, opt, mbs[rank * 2:(rank + 1) * 2], dp=dp)
	torch.cuda.synchronize()
	out[rank] = (model.flat_parameters().detach().cpu(), calls)
	dist.barrier()
	dist.destroy_process_group()


def test_two_ranks_match_single_process():
	world, port = 2, _free_port()
	with mp.Manager() as mgr:
		out = mgr.dict()
		mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
		res = dict(out)
	assert torch.equal(res[0][0], res[1][0])
	assert len(res[0][1]) == 2 * 2 and res[0][1][0][0] > res[0][1][1][0]  # two layers per step, last layer first
	T, model, opt, mbs = _setup(seed=7)
	for _ in range(2):
		T.train_step(model, opt, mbs)
	torch.cuda.synchronize()
	torch.testing.assert_close(res[0][0], model.flat_parameters().detach().cpu(), atol=2e-5, rtol=1e-4)
