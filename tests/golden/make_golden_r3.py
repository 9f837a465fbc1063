#!/usr/bin/env python3
"""Round-3 golden fixture generated with the reference's own decoder module and torch's own optimizer / schedulers (build container only: needs /root/reference).

    python tests/golden/make_golden_r3.py

``ref_train_resume.pt``: the training state a `.train` file of the REFERENCE carries (train.py:1450-1473) -- `model_state_dict` of the reference's PrefixedIterDecoder,
`optimizer_type` / `optimizer_state_dict` of `torch.optim.AdamW` over the reference's parameter groups (train.py:1103-1119: < 2-D tensors without weight decay first,
then the >= 2-D ones), `scheduler_warmup_state_dict` (LinearLR) and `scheduler_state_dict` (CosineAnnealingLR) as set up at :1138-1158 and stepped once per chunk
(:1339-1342) -- taken after a few optimizer steps, plus what the reference does NEXT from that state: the micro-batches of the following optimizer steps, their losses,
gradient norms, learning rates, and the weights afterwards.  train.py itself is not importable here; the loop below restates :1252-1286, :1339-1342 around the
imported decoder, as make_golden.py's train_case does."""
import dataclasses
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
from make_golden import O, check  # noqa: E402


def main():
	spec, seed = MG.SMALL, 1300
	accum, lr, warm, max_chunks, final_lr = 2, 2e-3, 2, 6, 0.0
	model, sd, _ = MG.ref_model(spec, seed)
	model.train()
	params = [p for p in model.parameters() if p.requires_grad]
	groups = [{"params": [p for p in params if p.dim() < 2], "weight_decay": 0.0}, {"params": [p for p in params if p.dim() >= 2], "weight_decay": 0.1}]
	opt = torch.optim.AdamW(groups, lr=lr, betas=(0.9, 0.95), weight_decay=0.1)
	warmup = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1 / (warm + 1), end_factor=1, total_iters=warm)
	cosine = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=max(max_chunks + 1 - 0, 1), eta_min=final_lr)

	def one_step(step):
		"""one optimizer step = one chunk here: accum micro-batches, clip, AdamW, then both schedulers (train.py:1252-1286, :1339-1342)"""
		mbs = [MG.synth_batch(spec, 8, seed * 10 + step * accum + j) for j in range(accum)]
		opt.zero_grad(set_to_none=True)
		total = 0.0
		for embed, target, pad, weight in mbs:
			out = model(embed=embed, target=target, target_padding=pad, target_weight=weight, calc_loss=True, calc_correct=True, only_pred=False, guide_targets=None)
			loss = out[2] / out[3] / accum
			loss.backward()
			total += float(loss)
		gn = float(torch.nn.utils.clip_grad_norm_(params, max_norm=1.0, error_if_nonfinite=True))
		used_lr = opt.param_groups[0]["lr"]
		opt.step()
		warmup.step()
		cosine.step()
		return mbs, total, gn, used_lr

	for step in range(1, 4):
		one_step(step)
	clone = lambda x: x.clone() if torch.is_tensor(x) else x
	opt_sd = opt.state_dict()
	checkpoint = dict(
		model_state_dict={k: v.detach().clone() for k, v in model.state_dict().items()},
		optimizer_type=f"{type(opt).__module__}.{type(opt).__qualname__}",
		optimizer_state_dict=dict(state={i: {k: clone(v) for k, v in st.items()} for i, st in opt_sd["state"].items()}, param_groups=[dict(g) for g in opt_sd["param_groups"]]),
		scheduler_warmup_state_dict=dict(warmup.state_dict()), scheduler_state_dict=dict(cosine.state_dict()),
		cfg_flat=dict(init_lr=lr, final_lr=final_lr, lr_scheduler="cosine", lr_warmup=warm, beta1=0.9, beta2=0.95, weight_decay=0.1, weight_decay_1d=False, gradient_clip=1.0,
		              accum_factor=accum, max_chunks=max_chunks),
		train_loop_state=dict(chunk_id=3),
	)
	assert checkpoint["optimizer_type"] == "torch.optim.adamw.AdamW"
	# the oracle's AdamW restatement picks the state up too (cross-check of the mapping rule before anything is written)
	order = [k for k in checkpoint["model_state_dict"] if k != "causality_mask"]
	order = [k for k in order if checkpoint["model_state_dict"][k].dim() < 2] + [k for k in order if checkpoint["model_state_dict"][k].dim() >= 2]
	ids = [i for g in opt_sd["param_groups"] for i in g["params"]]
	my_params = {k: checkpoint["model_state_dict"][k].clone() for k in order}
	my_state = {k: (opt_sd["state"][i]["exp_avg"].clone(), opt_sd["state"][i]["exp_avg_sq"].clone()) for i, k in zip(ids, order)}
	nxt = []
	for step in range(4, 6):
		mbs, total, gn, used_lr = one_step(step)
		req = {k: v.clone().requires_grad_(True) for k, v in my_params.items()}
		mine, _ = O.loss_for_step(dict(req, causality_mask=sd["causality_mask"]), spec, mbs)
		mine.backward()
		my_gn = O.clip_and_adamw(my_params, {k: v.grad for k, v in req.items()}, my_state, step, used_lr)
		assert abs(float(mine) - total) <= 1e-5 and abs(float(my_gn) - gn) <= 1e-4 * max(1.0, gn), (float(mine), total, float(my_gn), gn)
		nxt.append(dict(batches=mbs, loss=total, grad_norm=gn, lr=used_lr, lr_after=opt.param_groups[0]["lr"],
		                weights={k: v.detach().clone() for k, v in model.state_dict().items() if k != "causality_mask"}))
	for k, v in nxt[-1]["weights"].items():
		check(f"resume.final.{k}", v, my_params[k], atol=2e-5, rtol=1e-4)
	out = dict(spec=dataclasses.asdict(spec), seed=seed, accum=accum, checkpoint=checkpoint, next=nxt)
	path = os.path.join(HERE, "ref_train_resume.pt")
	torch.save(out, path)
	print(f"wrote ref_train_resume.pt: {os.path.getsize(path) / 1024:.1f} KiB; lrs {[n['lr'] for n in nxt]}, losses {[round(n['loss'], 4) for n in nxt]}")


if __name__ == "__main__":
	main()
