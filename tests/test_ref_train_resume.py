"""Resuming from a `.train` file the REFERENCE wrote (missing #4): torch.optim.AdamW's per-parameter state in the reference's parameter-group order
(train.py:1103-1119) and torch's LinearLR / CosineAnnealingLR state dicts (:1138-1158) map into FusedAdamW's flat moments and ChunkSchedule.

tests/golden/ref_train_resume.pt (tests/golden/make_golden_r3.py): state after three optimizer steps of the reference decoder under torch.optim.AdamW + both schedulers,
and what the reference does next -- two more steps: micro-batches, losses, gradient norms, learning rates, weights."""
import pytest
import torch

from conftest import load_golden
from helpers import make_decoder
from oracle import decoder_oracle as O

FX = load_golden("ref_train_resume.pt")
SPEC = O.DecoderSpec(**FX["spec"])
CK = FX["checkpoint"]


def _resumed(device):
	from novic_amd import train as T
	model, _ = make_decoder(SPEC, sd=CK["model_state_dict"], device=device)
	model.eval()  # the fixture trained without dropout
	cf = CK["cfg_flat"]
	opt = T.FusedAdamW(model, lr=cf["init_lr"], betas=(cf["beta1"], cf["beta2"]), weight_decay=cf["weight_decay"], max_norm=cf["gradient_clip"], weight_decay_1d=cf["weight_decay_1d"])
	names = dict(model.named_parameters())
	order = [k for k in CK["model_state_dict"] if k in names]
	opt.load_reference_state_dict(CK["optimizer_state_dict"], order)
	sched = T.ChunkSchedule(opt, cf["init_lr"], cf["lr_warmup"], cf["lr_scheduler"], max(cf["max_chunks"] + 1 - CK["train_loop_state"]["chunk_id"], 1), cf["final_lr"])
	sched.load_reference_state_dicts(CK["scheduler_warmup_state_dict"], CK["scheduler_state_dict"])
	return model, opt, sched, order


def test_reference_optimizer_and_scheduler_state_land_in_the_right_slots():
	model, opt, sched, order = _resumed("cpu")
	sd, osd = CK["model_state_dict"], CK["optimizer_state_dict"]
	assert opt.step_count == 3 and opt.param_groups[0]["betas"] == (0.9, 0.95) and opt.param_groups[0]["weight_decay"] == 0.1 and not opt.weight_decay_1d
	# torch's numbering: group 0 = the < 2-D tensors in model.parameters() order, group 1 = the others
	ref_order = [k for k in order if sd[k].dim() < 2] + [k for k in order if sd[k].dim() >= 2]
	ids = [i for g in osd["param_groups"] for i in g["params"]]
	assert len(ids) == len(ref_order) == len(osd["state"])
	for i, name in zip(ids, ref_order):
		o, shape = model._offsets[name]
		n = sd[name].numel()
		assert tuple(shape) == tuple(sd[name].shape)
		assert torch.equal(opt.exp_avg[o:o + n].view(shape), osd["state"][i]["exp_avg"]) and torch.equal(opt.exp_avg_sq[o:o + n].view(shape), osd["state"][i]["exp_avg_sq"])
		assert float(osd["state"][i]["exp_avg"].abs().max()) > 0
	# every element of the flat moments that belongs to a parameter was written (nothing else is non-zero)
	assert int((opt.exp_avg != 0).sum()) == sum(int((osd["state"][i]["exp_avg"] != 0).sum()) for i in ids)
	# the schedule continues on torch's learning-rate sequence
	assert sched.chunks_done == 3 and opt.lr == pytest.approx(FX["next"][0]["lr"], rel=1e-12)
	for nx in FX["next"]:
		assert opt.lr == pytest.approx(nx["lr"], rel=1e-9)
		sched.step()
		assert opt.lr == pytest.approx(nx["lr_after"], rel=1e-9)
	# a state dict in another group layout is refused rather than mis-assigned
	bad = dict(osd, param_groups=[dict(osd["param_groups"][0], params=osd["param_groups"][0]["params"][:-1]), osd["param_groups"][1]])
	with pytest.raises(ValueError):
		opt.load_reference_state_dict(bad, order)


@pytest.mark.gpu
def test_next_steps_after_a_reference_written_state_match_the_reference():
	from novic_amd import train as T
	model, opt, sched, _ = _resumed("cuda")
	fresh, _ = make_decoder(SPEC, sd=CK["model_state_dict"], device="cuda")  # same weights, optimizer state dropped (what round 2 did)
	fresh.eval()
	cf = CK["cfg_flat"]
	fopt = T.FusedAdamW(fresh, lr=FX["next"][0]["lr"], betas=(cf["beta1"], cf["beta2"]), weight_decay=cf["weight_decay"], max_norm=cf["gradient_clip"])
	for nx in FX["next"]:
		lr = opt.lr
		assert lr == pytest.approx(nx["lr"], rel=1e-9)
		mbs = [tuple(None if t is None else t.cuda() for t in mb) for mb in nx["batches"]]
		stats, gnorm = T.train_step(model, opt, mbs)
		fopt.param_groups[0]["lr"] = lr
		T.train_step(fresh, fopt, mbs)
		torch.cuda.synchronize()
		assert abs(float((stats[1] / stats[0]).mean()) - nx["loss"]) <= 1e-2 * nx["loss"]
		assert abs(float(gnorm) - nx["grad_norm"]) <= 3e-2 * nx["grad_norm"]
		worst = worst_fresh = 0.0
		for k, p in model.named_parameters():
			d = (p.detach().cpu() - nx["weights"][k]).abs()
			worst = max(worst, float(d.max()))
			# with the moments loaded the update is a smooth function of the gradient: a bf16 gradient error of a few percent moves a weight by a fraction of lr
			assert float(d.mean()) <= 0.03 * lr, (k, float(d.mean()), lr)
			worst_fresh = max(worst_fresh, float((dict(fresh.named_parameters())[k].detach().cpu() - nx["weights"][k]).abs().max()))
		assert worst <= 0.35 * lr, (worst, lr)
		assert worst_fresh >= 3 * worst  # the comparison has teeth: restarting the moments lands far away (first AdamW step = lr * sign(g))
		sched.step()
		assert opt.lr == pytest.approx(nx["lr_after"], rel=1e-9)
