"""GPU parity of the fused noise kernel (injected randoms: value-for-value vs the golden reference outputs; Philox mode: statistics) and of
the fused clip + AdamW kernels against the oracle's restatement of clip_grad_norm_ + torch.optim.AdamW."""
import math

import pytest
import torch

from conftest import load_golden
from oracle import decoder_oracle as O
from oracle import noise_oracle as NO

pytestmark = pytest.mark.gpu
NOISE = {c["name"]: c for c in load_golden("noise.pt")}


def _dev(*ts):
	return [None if t is None else t.cuda().contiguous() for t in ts]


def test_noise_injected_matches_reference_outputs():
	from novic_amd import ops
	rad = math.radians
	c = NOISE["gauss_elem"]
	e, z = _dev(c["embed"].clone(), c["z"])
	ops.noise_fused(e, ops.NOISE_GAUSS_ELEM, vec_norm=c["vec_norm"], inj_z1=z)
	torch.testing.assert_close(e.cpu(), c["out"], atol=2e-6, rtol=1e-5)
	c = NOISE["gauss_vec"]
	e, z, r = _dev(c["embed"].clone(), c["z"], c["r"].reshape(-1))
	ops.noise_fused(e, ops.NOISE_GAUSS_VEC, vec_norm=c["vec_norm"], inj_z1=z, inj_row=r)
	torch.testing.assert_close(e.cpu(), c["out"], atol=2e-6, rtol=1e-5)
	c = NOISE["uniform_angle"]
	u = ((c["angle"] - rad(c["angle_min"])) / (rad(c["angle_max"]) - rad(c["angle_min"]))).reshape(-1)
	e, z, u = _dev(c["embed"].clone(), c["z"], u)
	ops.noise_fused(e, ops.NOISE_UNIFORM_ANGLE, angle_min=rad(c["angle_min"]), angle_max=rad(c["angle_max"]), inj_z1=z, inj_row=u)
	torch.testing.assert_close(e.cpu(), c["out"], atol=3e-6, rtol=1e-5)
	c = NOISE["gauss_angle"]
	e, z, r = _dev(c["embed"].clone(), c["z"], c["r"].reshape(-1))
	ops.noise_fused(e, ops.NOISE_GAUSS_ANGLE, angle_std=rad(c["angle_std"]), angle_max=rad(c["angle_max"]), inj_z1=z, inj_row=r)
	torch.testing.assert_close(e.cpu(), c["out"], atol=3e-6, rtol=1e-5)
	c = NOISE["gauss_elem_uniform_angle"]
	e, zg, za, ua, um = _dev(c["embed"].clone(), c["z_gauss"], c["z_angle"], c["u_angle"].reshape(-1), c["u_mix"].reshape(-1))
	ops.noise_fused(e, ops.NOISE_GAUSS_ELEM_UNIFORM_ANGLE, vec_norm=c["vec_norm"], angle_min=rad(c["angle_min"]), angle_max=rad(c["angle_max"]), mix_ratio=c["mix_ratio"],
	                inj_z1=zg, inj_z2=za, inj_row=ua, inj_mix=um)
	torch.testing.assert_close(e.cpu(), c["out"], atol=3e-6, rtol=1e-5)
	assert 0 < int((c["u_mix"] < c["mix_ratio"]).sum()) < c["embed"].shape[0]  # both branches exercised
	c = NOISE["mean_shift"]
	e, sh = _dev(c["embed"].clone(), c["shift"].reshape(-1))
	ops.noise_fused(e, ops.NOISE_NONE, mean_shift=sh)
	torch.testing.assert_close(e.cpu(), c["out"], atol=1e-6, rtol=1e-5)


def test_noise_philox_statistics():
	"""Module surface + in-kernel RNG: unit rows, E||noise|| = vec_norm for GaussElem, rotation angle uniform in [min, max], mix fraction."""
	from novic_amd import embedding_noise as EN
	B, F = 8192, 512
	g = torch.Generator().manual_seed(0)
	base = torch.nn.functional.normalize(torch.randn(B, F, generator=g), dim=-1).cuda()
	mod = EN.EmbeddingNoise.create("GaussElem", F, 3.25, 0, 0, 0, 0)
	out = mod(base.clone())
	assert torch.allclose(out.norm(dim=1), torch.ones(B, device="cuda"), atol=1e-5)
	# out ~ (e + n)/|e + n| with |n| ~ 3.25: cos(angle) = (1 + e.n)/|e+n| ~ 1/sqrt(1 + 3.25^2)
	cos = (out * base).sum(dim=1)
	assert abs(float(cos.mean()) - 1 / math.sqrt(1 + 3.25 ** 2)) < 0.01
	out2 = mod(base.clone())
	assert not torch.equal(out, out2)  # a fresh Philox stream per call
	mod = EN.EmbeddingNoise.create("UniformAngle", F, 0, 45.0, 75.0, 0, 0)
	ang = torch.rad2deg(torch.acos((mod(base.clone()) * base).sum(dim=1).clamp(-1, 1)))
	assert float(ang.min()) >= 44.99 and float(ang.max()) <= 75.01
	assert abs(float(ang.mean()) - 60.0) < 0.5 and abs(float(ang.std()) - 30 / math.sqrt(12)) < 0.3
	mod = EN.EmbeddingNoise.create("GaussElemUniformAngle", F, 3.25, 45.0, 75.0, 0, 0.15)
	ang = torch.rad2deg(torch.acos((mod(base.clone()) * base).sum(dim=1).clamp(-1, 1)))
	frac_rot = float(((ang >= 44.99) & (ang <= 75.01)).float().mean())  # GaussElem rows sit near acos(0.294) = 72.9 deg too, so bound from both sides
	gauss_in_band = 1.0  # nearly all Gaussian rows also land in the band at F = 512; check the mixture through its mean angle instead
	assert abs(float(ang.mean()) - (0.15 * 60.0 + 0.85 * math.degrees(math.acos(1 / math.sqrt(1 + 3.25 ** 2))))) < 0.6
	mod = EN.EmbeddingNoise.create("GaussAngle", F, 0, 0, 40.0, 25.0, 0)
	ang = torch.rad2deg(torch.acos((mod(base.clone()) * base).sum(dim=1).clamp(-1, 1)))
	assert float(ang.max()) <= 40.01 and abs(float((ang >= 39.99).float().mean()) - 2 * (1 - 0.5 * (1 + math.erf(40 / 25 / math.sqrt(2))))) < 0.02
	mod = EN.EmbeddingNoise.create("GaussVec", F, 0.7, 0, 0, 0, 0)
	out = mod(base.clone())
	assert torch.allclose(out.norm(dim=1), torch.ones(B, device="cuda"), atol=1e-5)
	assert EN.EmbeddingNoise.create("", F, 0, 0, 0, 0, 0) is None
	with pytest.raises(ValueError):
		EN.EmbeddingNoise.create("nope", F, 0, 0, 0, 0, 0)


def test_fused_clip_adamw_matches_oracle():
	from helpers import make_decoder
	from novic_amd import train as T
	spec = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)
	model, sd = make_decoder(spec, seed=21, device="cuda")
	opt = T.FusedAdamW(model, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	params = {k: v.clone() for k, v in sd.items() if k != "causality_mask"}
	state = {}
	g = torch.Generator().manual_seed(3)
	for step in range(1, 4):
		grads = {k: torch.randn(v.shape, generator=g) * (0.5 if step == 1 else 0.02) for k, v in params.items()}  # step 1 clips, later steps do not
		model.flat_grad().zero_()
		for k, p in model.named_parameters():
			p.grad.copy_(grads[k].cuda())
		norm = opt.step()
		ref_norm = O.clip_and_adamw(params, grads, state, step, 2e-3)
		assert abs(float(norm) - float(ref_norm)) <= 1e-5 * float(ref_norm)
		for k, p in model.named_parameters():
			torch.testing.assert_close(p.detach().cpu(), params[k], atol=2e-6, rtol=2e-5)
		# the bf16 shadow the GEMMs read follows the master
		o, shape = model._offsets["logits_linear.weight"]
		assert torch.equal(model._flat16[o:o + math.prod(shape)].view(shape).float().cpu(), params["logits_linear.weight"].to(torch.bfloat16).float())


def test_fused_adamw_steps_without_host_syncs_match_oracle():
	"""the optimizer's hyper-parameters (lr, bias corrections) must belong to the step that launched them even when the host runs many
	steps ahead of the device -- they travel by value in the kernel arguments.  Ten steps with a changing learning rate, gradients resident on the
	device, no synchronisation until the end; behind a long-running kernel so the device really is behind the host."""
	from helpers import make_decoder
	from novic_amd import train as T
	spec = O.DecoderSpec(embed_dim=32, vocab_size=53, token_length=6, hidden_dim=64, feedfwd_dim=16, num_layers=2, num_heads=4)
	model, sd = make_decoder(spec, seed=22, device="cuda")
	opt = T.FusedAdamW(model, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	params = {k: v.clone() for k, v in sd.items() if k != "causality_mask"}
	names = [k for k, _ in model.named_parameters()]
	g = torch.Generator().manual_seed(4)
	steps = 10
	grads = [{k: torch.randn(params[k].shape, generator=g) * 0.05 for k in names} for _ in range(steps)]
	flat_grads = []
	for gs in grads:
		model.flat_grad().zero_()
		for k, p in model.named_parameters():
			p.grad.copy_(gs[k].cuda())
		flat_grads.append(model.flat_grad().clone())
	lrs = [2e-3 * (0.5 + 0.1 * i) for i in range(steps)]
	stall = torch.randn(4096, 4096, device="cuda")
	torch.cuda.synchronize()
	for _ in range(40):  # ~ tens of ms of queued work: every optimizer launch below is enqueued before the first one runs
		stall = (stall @ stall).clamp_(-1, 1)
	for i in range(steps):
		opt.param_groups[0]["lr"] = lrs[i]
		model.flat_grad().copy_(flat_grads[i])
		opt.step()
	torch.cuda.synchronize()
	state = {}
	for i in range(steps):
		O.clip_and_adamw(params, grads[i], state, i + 1, lrs[i])
	for k, p in model.named_parameters():
		torch.testing.assert_close(p.detach().cpu(), params[k], atol=5e-6, rtol=5e-5)
