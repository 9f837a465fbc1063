"""training_loop on the GPU: chunk accounting, EWA metrics replay, schedule stepping, checkpoint round trip (resume gives identical weights),
and equivalence of the merged optimizer step with per-micro-batch accumulation (the reference's semantics)."""
import dataclasses
import os

import pytest
import torch

from helpers import make_decoder, synth_batch, to_dev
from oracle import decoder_oracle as O

pytestmark = pytest.mark.gpu
SPEC = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)


class Loader:
	def __init__(self, n, B):
		self.batches = [to_dev(*synth_batch(SPEC, B, seed=1000 + i, max_len=4)) for i in range(n)]
		# fixed C across batches so an optimizer step can merge its micro-batches
		C = max(b[1].shape[1] for b in self.batches)
		self.batches = [(e, torch.nn.functional.pad(t, (0, C - t.shape[1])), torch.nn.functional.pad(m, (0, C - m.shape[1]), value=True), w) for e, t, m, w in self.batches]

	def __len__(self):
		return len(self.batches)

	def __iter__(self):
		return iter([(e.clone(), t, m, w) for e, t, m, w in self.batches])


def test_merged_step_equals_accumulated_micro_batches():
	from novic_amd import train as T
	mbs = Loader(4, 16).batches
	a, sd = make_decoder(SPEC, seed=7, device="cuda")
	b, _ = make_decoder(SPEC, seed=7, device="cuda")
	a.eval(); b.eval()
	oa, ob = T.FusedAdamW(a, lr=1e-3), T.FusedAdamW(b, lr=1e-3)
	sa, na = T.train_step(a, oa, mbs, merged=True)
	sb, nb = T.train_step(b, ob, mbs, merged=False)
	torch.testing.assert_close(sa, sb, rtol=1e-4, atol=1e-4)
	assert abs(float(na) - float(nb)) <= 2e-3 * float(nb)
	assert float((a.flat_grad() - b.flat_grad()).norm() / b.flat_grad().norm()) < 5e-3
	assert float((a.flat_parameters() - b.flat_parameters()).abs().max()) < 2e-3
	# and against the oracle's restatement of one reference optimizer step (fp32): loss exact-ish, update direction equal
	params = {k: v.clone() for k, v in sd.items() if k != "causality_mask"}
	req = {k: v.clone().requires_grad_(True) for k, v in params.items()}
	cpu_mbs = [tuple(None if t is None else t.cpu() for t in mb) for mb in mbs]
	total, _ = O.loss_for_step(dict(req, causality_mask=sd["causality_mask"]), SPEC, cpu_mbs)
	total.backward()
	gn = O.clip_and_adamw(params, {k: v.grad for k, v in req.items()}, {}, 1, 1e-3)
	assert abs(float((sa[1] / sa[0]).mean()) - float(total)) <= 1e-2 * float(total)
	assert abs(float(na) - float(gn)) <= 3e-2 * float(gn)
	for k, p in a.named_parameters():  # accumulated (pre-clip) gradients against the fp32 oracle, per tensor
		ref = req[k].grad
		assert float((p.grad.cpu() - ref).norm() / (ref.norm() + 1e-12)) <= 6e-2, k


def test_training_loop_chunks_checkpoint_and_resume(tmp_path):
	from novic_amd import embedding_dataset, embedding_noise, train as T
	loader = Loader(12, 16)
	info = embedding_dataset.LoaderInfo(num_workers=0, prefetch_factor=0, pin_memory=False, on_device=True, batch_size=16, batch_size_last=0, complete_batches=12,
	                                    incomplete_batch=False, epoch_batches=12, epoch_samples=192, available_samples=192)
	cfg_flat = dict(model="PrefixedIterDecoder", note="test")

	def run(model, max_chunks, state=None, opt_state=None, sched_state=None):
		ga = embedding_dataset.GradAccum(loader, info, accum_size=2, drop_last=True)
		C = T.make_train_loop_config(run_dir=str(tmp_path), batch_size=16, epoch_batches=ga.loader_batches, num_valid_targets=2, accum_size=2, chunk_scale=32,
		                             max_chunks=max_chunks, max_epochs=0, save_every_min=1, save_every_max=2, save_top1_min=0.0)
		assert C.chunk_batches == 4 and C.chunk_samples == 64
		S = state or T.TrainLoopState()
		opt = T.FusedAdamW(model, lr=2e-3)
		sched = T.ChunkSchedule(opt, 2e-3, 1, "cosine", max_chunks + 1 - S.chunk_id if state else max_chunks, 0.0)
		if opt_state:
			opt.load_state_dict(opt_state)
		if sched_state:
			sched.load_state_dict(sched_state)
		noise = embedding_noise.EmbeddingNoise.create("GaussElem", SPEC.embed_dim, 0.5, 0, 0, 0, 0)
		infos = []
		T.training_loop(cfg_flat, C, S, model, ("",) + tuple(f"n{i}" for i in range(5)), 1, None, noise, ga, opt, sched, torch.device("cuda"), log=lambda m: None,
		                on_chunk=infos.append)
		return C, S, opt, sched, infos

	model, _ = make_decoder(SPEC, seed=11, dropout=0.1, device="cuda")
	C, S, opt, sched, infos = run(model, 5)
	assert S.chunk_id == 5 and S.batch_id == 20 and S.sample_id == 320 and S.epoch_id == 2
	assert len(infos) == 5 and infos[-1]["loss"] < infos[0]["loss"] and all(i["samples_per_s"] > 0 for i in infos)
	assert 0 <= S.ewa_train_top1 <= 1 and S.ewa_train_loss > 0 and S.saved_num >= 2
	files = sorted(f for f in os.listdir(tmp_path) if f.endswith(".train"))
	assert files, "no checkpoint written"
	ckpt = torch.load(os.path.join(tmp_path, files[-1]), weights_only=False)
	assert set(ckpt) >= {"cfg_flat", "target_config", "data_config", "model_state_dict", "target_nouns", "num_invalid_target_nouns", "train_loop_config", "train_loop_state",
	                     "optimizer_type", "optimizer_state_dict", "scheduler_state_dict", "amp_scaler_enabled"}
	assert set(ckpt["model_state_dict"]) == set(model.state_dict())
	# the loop state is saved AFTER the chunk counter moved on (reference train.py:1374-1389): the next chunk to train
	assert ckpt["train_loop_state"]["saved_chunk_id"] == S.saved_chunk_id and ckpt["train_loop_state"]["chunk_id"] in (S.saved_chunk_id, S.saved_chunk_id + 1)
	# a fresh model loads the checkpoint strictly and reproduces the saved weights
	fresh, _ = make_decoder(SPEC, seed=99, dropout=0.1, device="cuda")
	fresh.load_state_dict(ckpt["model_state_dict"], strict=True)
	for k, v in ckpt["model_state_dict"].items():
		assert torch.equal(fresh.state_dict()[k].cpu(), v)


def test_full_size_step_is_the_sum_of_its_micro_batches():
	"""The BASELINE configuration at full size (6 layers, d = 512, V = 6912, 16 micro-batches of 512 = 8192 samples per optimizer step) through a property
	that needs no oracle: the merged step (one packed forward/backward over all 8192 samples, what bench.py times) must give the loss statistics of
	the 16 separate micro-batch passes and their accumulated gradient (dropout off; bf16 GEMM operands, fp32 sums in another order), and the packed
	layout must process exactly the rows that are not padding."""
	from novic_amd import train as T
	spec = O.DecoderSpec(embed_dim=512, vocab_size=6912, token_length=12)
	mbs = []
	for i in range(16):
		e, t, m, w = synth_batch(spec, 512, seed=500 + i, max_len=6)
		C = 7
		mbs.append(to_dev(e, torch.nn.functional.pad(t, (0, C - t.shape[1])), torch.nn.functional.pad(m, (0, C - m.shape[1]), value=True), w))
	a, _ = make_decoder(spec, seed=7, device="cuda")
	b, _ = make_decoder(spec, seed=7, device="cuda")
	a.eval(); b.eval()
	oa, ob = T.FusedAdamW(a, lr=1e-3), T.FusedAdamW(b, lr=1e-3)
	from novic_amd import ops
	ops.gemm_tile_counts(reset=True)
	sa, na = T.train_step(a, oa, mbs, merged=True)
	counts = ops.gemm_tile_counts()
	# the merged step runs its big GEMMs on the 256-wide tile, and the logits input gradient (a device row count over 448 allocated tiles) with its K-split tail planned on the device
	assert counts["t256"] >= 20 and counts["ksplit_tail_device"] >= 1, counts
	ga = a.flat_grad().clone()
	sb, nb = T.train_step(b, ob, mbs, merged=False)
	gb = b.flat_grad().clone()
	torch.cuda.synchronize()
	torch.testing.assert_close(sa, sb, rtol=2e-4, atol=1e-3)           # basis, loss sum, #correct, #tokens per micro-batch
	assert abs(float(na) - float(nb)) <= 2e-3 * float(nb)               # global gradient norm (what the clip uses)
	assert float((ga - gb).norm() / gb.norm()) < 5e-3
	# rows the packed layout kept: the prefix (4) + the label tokens that are INPUTS, i.e. every unpadded target position but the last one (END is
	# only ever predicted): embedding_decoder.py:696-712 pads input position P + c with target position c + 1
	kept = int(a._ws.bufs["train:seq_total"][0])
	want = sum(int((4 - 1 + (~m).sum(dim=1)).sum()) for _, _, m, _ in mbs)
	assert kept == want and kept < 8192 * 10
	assert int(a._ws.bufs["train:cmp_count"][0]) == sum(int((~m).sum()) for _, _, m, _ in mbs)


def test_step_over_a_loader_group_is_the_step_over_its_concatenated_micro_batches():
	"""train_step handed the GroupSlices of one loader group (embedding_cache.DeviceLoader(group = accum): the step's micro-batches assembled into one set of buffers) uses
	the buffers whole; the same micro-batches as plain tuples are concatenated first: the same statistics, gradient norm and gradients (same values, same
	launches); slices out of order or of two groups fall back to the concatenation."""
	from novic_amd import embedding_cache as EC, train as T
	spec = O.DecoderSpec(embed_dim=64, vocab_size=307, token_length=8)
	mbs = []
	for i in range(4):
		e, t, m, w = synth_batch(spec, 96, seed=40 + i, max_len=5)
		C = 6
		mbs.append(to_dev(e, torch.nn.functional.pad(t, (0, C - t.shape[1])), torch.nn.functional.pad(m, (0, C - m.shape[1]), value=True), w))
	full = tuple(None if mbs[0][k] is None else torch.cat([mb[k] for mb in mbs], dim=0) for k in range(4))
	B = mbs[0][0].shape[0]
	slices = [EC.GroupSlice(tuple(None if t is None else t[g * B:(g + 1) * B] for t in full), full, g, 4) for g in range(4)]
	assert T._group_of(slices) is full and T._group_of(mbs) is None and T._group_of(slices[::-1]) is None and T._group_of(slices[:3]) is None
	res = []
	for batches in (mbs, slices):
		a, _ = make_decoder(spec, seed=7, device="cuda")
		a.eval()
		oa = T.FusedAdamW(a, lr=1e-3)
		keep = [t.clone() for t in full if t is not None]
		st, nm = T.train_step(a, oa, batches, merged=True)
		res.append((st.clone(), nm.clone(), a.flat_grad().clone()))
		assert all(torch.equal(x, y) for x, y in zip(keep, [t for t in full if t is not None]))  # (no noise here: the step leaves the loader's buffers as they were)
	torch.cuda.synchronize()
	# (same values through the same launches; the small-shape weight gradients and the embedding gradient accumulate with fp32 atomics, so two runs of ONE path already differ in
	# the last bits)
	torch.testing.assert_close(res[0][0], res[1][0], rtol=1e-6, atol=1e-6)
	torch.testing.assert_close(res[0][1], res[1][1], rtol=1e-5, atol=0)
	assert float((res[0][2] - res[1][2]).norm() / res[0][2].norm()) < 1e-5


def test_gpu_replays_the_reference_training_trajectory():
	"""a17: the six optimizer steps of tests/golden/train_trajectory.pt (reference decoder + torch.optim.AdamW + clip_grad_norm_, accum 2, dropout 0) through
	train.train_step on the GPU -- with NO host synchronisation between the steps, the way training_loop drives it (one sync per chunk): per-step
	losses and gradient norms within the bf16 tolerance, final weights against the reference's samples."""
	from conftest import load_golden
	from novic_amd import train as T
	tr = load_golden("train_trajectory.pt")
	spec = O.DecoderSpec(**tr["spec"])
	model, _ = make_decoder(spec, seed=tr["seed"], device="cuda")
	model.eval()  # dropout off, gradients on (forward_backward keeps activations either way)
	opt = T.FusedAdamW(model, lr=tr["lr"], betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	steps = [[to_dev(*mb) for mb in mbs] for mbs in tr["batches"]]
	torch.cuda.synchronize()
	stats, norms = [], []
	for mbs in steps:  # nothing below reads a device value back
		st, gn = T.train_step(model, opt, mbs, merged=False)
		stats.append(st.clone()); norms.append(gn.clone())
	torch.cuda.synchronize()
	for i, (st, gn) in enumerate(zip(stats, norms)):
		loss = float((st[1] / st[0]).mean())
		assert abs(loss - tr["losses"][i]) <= 1e-2 * abs(tr["losses"][i]), (i, loss, tr["losses"][i])
		assert abs(float(gn) - tr["grad_norms"][i]) <= 5e-2 * max(1.0, tr["grad_norms"][i]), (i, float(gn), tr["grad_norms"][i])
	# final weights against the reference's samples: AdamW normalises every element's move to ~lr per step whatever the gradient's size, so an element
	# whose gradient is smaller than the bf16 GEMM noise may walk the other way -- bounded by 2 * steps * lr; the update as a whole must point the same way
	init = O.init_state_dict(spec, seed=tr["seed"])
	d_gpu, d_ref = [], []
	for k, p in model.named_parameters():
		pick = lambda t: t.detach().cpu().flatten()[:: max(1, t.numel() // 32)][:32]
		got, ref, w0 = pick(p), tr["final_samples"][k], pick(init[k])
		assert float((got - ref).abs().max()) <= 2 * len(steps) * tr["lr"] * 1.05 + 1e-6, k
		d_gpu.append(got - w0); d_ref.append(ref - w0)
	d_gpu, d_ref = torch.cat(d_gpu), torch.cat(d_ref)
	cos = float(torch.dot(d_gpu, d_ref) / (d_gpu.norm() * d_ref.norm()))
	assert cos >= 0.97, cos
	assert float((d_gpu - d_ref).abs().mean()) <= 0.5 * tr["lr"]


def test_parameter_written_through_torch_after_cuda_reaches_the_kernels():
	"""after .cuda() every Parameter has its own version counter; an in-place write through a parameter must still invalidate the bf16 shadow
	(and the transposed shadows) the GEMMs read."""
	model, _ = make_decoder(SPEC, seed=5)
	model = model.cuda()
	model.eval()
	embed, target, pad, weight = to_dev(*synth_batch(SPEC, 8, seed=1))
	with torch.no_grad():
		before = model(embed, target, pad, weight, True, True, False, None)[0].clone()
		model.transformer.norm.weight.mul_(0.5)       # final LayerNorm gain: logits scale by exactly 0.5 (up to bf16 rounding of xf)
		after = model(embed, target, pad, weight, True, True, False, None)[0].clone()
		model.logits_linear.weight.copy_(torch.zeros_like(model.logits_linear.weight))
		zero = model(embed, target, pad, weight, True, True, False, None)[0]
	torch.testing.assert_close(after, 0.5 * before, atol=2e-2 * float(before.abs().max()), rtol=0)
	assert float(before.abs().max()) > 0.1 and float(zero.abs().max()) == 0.0
	# torch.optim on the parameters (the reference loop's optimizer) is seen as well
	model.train()
	sgd = torch.optim.SGD(model.parameters(), lr=0.5)
	out = model(embed, target, pad, weight, True, False, False, None)
	(out[2] / out[3]).backward()
	v0 = model._flat_version()
	sgd.step()
	assert model._flat_version() != v0
