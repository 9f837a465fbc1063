"""The fused decode-layer kernels (csrc/decode_fused.hip) against the unfused kernel chain they replace: same MFMA order, the same bf16 rounding
points and the same LayerNorm operation sequence (csrc/common.hpp: products that an addition follows are kept apart from it in EVERY compilation), so the
outputs agree bit for bit; and against torch fp32 within bf16 tolerance.  Then end to end: decoding with and without the fused path agrees."""
import pytest
import torch

from helpers import make_decoder
from oracle import decoder_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,E,Kf", [(256, 512, 128), (1000, 512, 128), (37, 64, 32), (16, 128, 64), (2560, 256, 256)])
def test_fused_layer_kernels_match_unfused_chain(M, E, Kf):
	from novic_amd import ops
	assert ops.decode_fused_supported(E, Kf)
	g = torch.Generator().manual_seed(M + E)
	r = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).cuda()
	x = r(M, E)
	g1, g2 = (1 + 0.1 * torch.randn(E, generator=g)).cuda(), (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	wqkv, wo = r(3 * E, E, scale=E ** -0.5).to(torch.bfloat16), r(E, E, scale=E ** -0.5).to(torch.bfloat16)
	w1, w2 = r(Kf, E, scale=E ** -0.5).to(torch.bfloat16), r(E, Kf, scale=Kf ** -0.5).to(torch.bfloat16)
	att = r(M, E).to(torch.bfloat16)
	# unfused chain
	ln = torch.empty(M, E, dtype=torch.bfloat16, device="cuda")
	qkv_ref = torch.empty(M, 3 * E, dtype=torch.bfloat16, device="cuda")
	ops.layernorm_fwd(x, g1, ln, M, E)
	ops.gemm(ln, wqkv, M, 3 * E, E, out=qkv_ref)
	xm, x_ref = torch.empty(M, E, device="cuda"), torch.empty(M, E, device="cuda")
	h = torch.empty(M, Kf, dtype=torch.bfloat16, device="cuda")
	ops.gemm(att, wo, M, E, E, kind=ops.EPI_RESID_F32, out=xm, resid=x)
	ops.layernorm_fwd(xm, g2, ln, M, E)
	ops.gemm(ln, w1, M, Kf, E, kind=ops.EPI_GELU_BF16, out=h)
	ops.gemm(h, w2, M, E, Kf, kind=ops.EPI_RESID_F32, out=x_ref, resid=xm)
	# fused
	qkv = torch.full((M, 3 * E), float("nan"), dtype=torch.bfloat16, device="cuda")
	ops.decode_ln_gemm(x, g1, wqkv, qkv, M, 3 * E, E)
	xm2, h2 = torch.full((M, E), float("nan"), device="cuda"), torch.full((M, Kf), float("nan"), dtype=torch.bfloat16, device="cuda")
	ops.decode_gemm_resid(att, wo, x, xm2, M, E, E)
	assert torch.equal(xm2, xm)                                             # no LayerNorm involved: bit-identical to the 128^2 GEMM + residual epilogue
	ops.decode_ln_gemm(xm2, g2, w1, h2, M, Kf, E, gelu=True)
	ln2, qkv3, h3 = torch.empty(M, E, dtype=torch.bfloat16, device="cuda"), torch.empty_like(qkv), torch.empty_like(h2)
	ops.layernorm_fwd(x, g1, ln2, M, E)
	ops.decode_gemm(ln2, wqkv, qkv3, M, 3 * E, E)                            # LayerNorm launch + small-tile GEMM: bit-identical to the 128^2 chain
	assert torch.equal(qkv3, qkv_ref)
	ops.layernorm_fwd(xm, g2, ln2, M, E)
	ops.decode_gemm(ln2, w1, h3, M, Kf, E, gelu=True)
	assert torch.equal(h3, h)
	x_new = xm2.clone()
	ops.decode_gemm_resid(h2, w2, x_new, x_new, M, E, Kf)                   # in place
	# LayerNorm as a GEMM prologue against LayerNorm as its own launch: since round 5 (csrc/common.hpp `unfused`) every compilation of the shared row arithmetic runs the
	# same IEEE operation sequence -- before, one accumulated the variance with fused multiply-adds and the other did not, and ~1 % of the rows differed by bf16 ulps
	assert torch.equal(qkv, qkv_ref) and torch.equal(h2, h) and torch.equal(x_new, x_ref)
	# torch fp32 restatement (bf16 rounding only at the operands): loose tolerance
	lnf = torch.nn.functional.layer_norm(x, (E,), g1, None, 1e-5)
	torch.testing.assert_close(qkv.float(), lnf.to(torch.bfloat16).float() @ wqkv.float().T, atol=6e-2, rtol=3e-2)
	xm_t = x + att.float() @ wo.float().T
	ff = torch.nn.functional.gelu(torch.nn.functional.layer_norm(xm_t, (E,), g2, None, 1e-5) @ w1.float().T) @ w2.float().T
	torch.testing.assert_close(x_new, xm_t + ff, atol=8e-2, rtol=3e-2)


@pytest.mark.parametrize("M,E", [(256, 512), (1000, 512), (37, 128), (16, 256), (2560, 512)])
def test_fused_feed_forward_launch_is_bit_identical_to_its_two_kernels(M, E):
	"""novic_decode_ffn (round 5): norm2 -> linear1 -> GELU -> linear2 -> residual as one launch, against novic_decode_ln_gemm(gelu) + novic_decode_gemm_resid on the same
	inputs -- the same LayerNorm sequence, MFMA order and bf16 rounding points: equal bit for bit; rows that do not fill the last 16-row block; not in place."""
	from novic_amd import ops
	Kf = 128
	assert ops.decode_ffn_supported(E, Kf) and not ops.decode_ffn_supported(E, 64) and not ops.decode_ffn_supported(64, 128)
	g = torch.Generator().manual_seed(M + E)
	x = torch.randn(M, E, generator=g).cuda()
	gamma = (1 + 0.1 * torch.randn(E, generator=g)).cuda()
	w1 = (torch.randn(Kf, E, generator=g) * E ** -0.5).to(torch.bfloat16).cuda()
	w2 = (torch.randn(E, Kf, generator=g) * Kf ** -0.5).to(torch.bfloat16).cuda()
	h = torch.empty(M, Kf, dtype=torch.bfloat16, device="cuda")
	want = torch.full((M, E), float("nan"), device="cuda")
	ops.decode_ln_gemm(x, gamma, w1, h, M, Kf, E, gelu=True)
	ops.decode_gemm_resid(h, w2, x, want, M, E, Kf)
	got = torch.full((M + 3, E), float("nan"), device="cuda")
	ops.decode_ffn(x, gamma, w1, w2, got, M, E, Kf)
	assert torch.equal(got[:M], want) and bool(torch.isnan(got[M:]).all())  # ... and nothing behind the last row is written
	ff = x + torch.nn.functional.gelu(torch.nn.functional.layer_norm(x, (E,), gamma, None, 1e-5) @ w1.float().T) @ w2.float().T
	torch.testing.assert_close(got[:M], ff, atol=8e-2, rtol=3e-2)


@pytest.mark.parametrize("beam", [False, True])
def test_decode_identical_with_and_without_fusion(beam):
	spec = O.DecoderSpec(embed_dim=512, vocab_size=6912, token_length=8)
	model, _ = make_decoder(spec, seed=11, device="cuda")
	model.eval()
	e = torch.nn.functional.normalize(torch.randn(40, 512, generator=torch.Generator().manual_seed(3)), dim=-1).cuda()
	outs = []
	for fused in (True, False):
		model.decode_fused = fused
		model.__dict__.pop("_decode_sessions", None)
		with torch.no_grad():
			o = model.generate_beam(e, 4, 1.0, 0.0, None, False, 0.0, None, False) if beam else model.generate(e, True, True, 1.0, 0.0, None, None, False)
		outs.append([t.clone() if torch.is_tensor(t) else t for t in o])
	ids_a, ids_b = outs[0][0], outs[1][0]
	assert ids_a.shape == ids_b.shape and torch.equal(ids_a, ids_b)  # (bit-identical layers since round 5: no near-tie can flip)
	sc_a, sc_b = (outs[0][2], outs[1][2]) if beam else (outs[0][5], outs[1][5])
	same = (ids_a == ids_b).flatten(1 if not beam else 2).all(dim=-1)
	torch.testing.assert_close(sc_a[same], sc_b[same], atol=2e-2, rtol=1e-2)


@pytest.mark.parametrize("beam", [False, True])
def test_next_step_inputs_from_the_selection_kernel(beam):
	"""Unguided steps: the greedy / beam selection kernel also writes the next step's input rows W_tok[token] + pos (and the beams' K/V origin table) instead of
	novic_decode_embed / novic_kv_origin_update launches -- the same fp32 additions and the same integers, so ids, scores and per-step logits are bit-identical with the
	option on and off, eagerly and through the captured graphs."""
	spec = O.DecoderSpec(embed_dim=512, vocab_size=6912, token_length=9)
	model, _ = make_decoder(spec, seed=13, device="cuda")
	model.eval()
	e = torch.nn.functional.normalize(torch.randn(48, 512, generator=torch.Generator().manual_seed(4)), dim=-1).cuda()
	outs = []
	cls = type(model)
	prev = cls.decode_embed_fused
	try:
		for fused in (True, False):
			cls.decode_embed_fused = fused
			model.__dict__.pop("_decode_sessions", None)
			with torch.no_grad():
				runs = [model.generate_beam(e, 4, 1.0, 0.5, None, False, 0.0, None, False) if beam else model.generate(e, True, True, 1.0, 0.0, None, None, False) for _ in range(3)]
			for r in runs[1:]:  # eager, capture, replay
				for x, y in zip(runs[0], r):
					assert (x is None and y is None) or torch.equal(x, y)
			outs.append([t.clone() if torch.is_tensor(t) else t for t in runs[0]])
	finally:
		cls.decode_embed_fused = prev
	for x, y in zip(outs[0], outs[1]):
		assert (x is None and y is None) or torch.equal(x, y)
