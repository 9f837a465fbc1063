"""Direct parity of the row kernels (novic_layernorm_fwd / novic_layernorm_bwd) against torch fp32 LayerNorm (bias-free, the decoder's layer_bias=False
norms of embedding_decoder.py:1289-1300) and its autograd, including the row-selection window the final norm uses (only the label positions of each
sequence reach the logits, embedding_decoder.py:690) and row widths that are not multiples of the 256-element lane chunks.
Tolerances: the forward output is bf16 (rel 2^-8); the backward works in fp32 on a bf16 upstream gradient: |err| <= 2e-5 * scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ln(x, gamma, eps=1e-5):
	return torch.nn.functional.layer_norm(x, (x.shape[-1],), gamma, None, eps)


@pytest.mark.parametrize("rows,E", [(1000, 512), (77, 320), (4099, 768), (5, 2048), (13, 4)])
def test_layernorm_forward(rows, E):
	from novic_amd import ops
	g = torch.Generator().manual_seed(rows + E)
	x = torch.randn(rows, E, generator=g) * 3 + 0.5
	gamma = torch.randn(E, generator=g)
	y = torch.empty(rows, E, dtype=torch.bfloat16, device="cuda")
	ops.layernorm_fwd(x.cuda(), gamma.cuda(), y, rows, E)
	want = _ln(x, gamma)
	err = (y.float().cpu() - want).abs()
	assert float((err - want.abs() * 2 ** -8).max()) <= 1e-6


@pytest.mark.parametrize("A,seq_in,seq_out,seq_off,E,has_dx,g_out,drop", [
	(700, 1, 1, 0, 512, True, True, 0.0), (700, 1, 1, 0, 512, False, True, 0.0), (123, 10, 7, 3, 512, False, True, 0.0), (123, 10, 7, 3, 512, True, False, 0.0),
	(31, 5, 2, 1, 320, True, True, 0.0), (1, 1, 1, 0, 4, True, True, 0.0), (9000, 1, 1, 0, 768, True, True, 0.0), (6000, 10, 7, 3, 512, True, True, 0.1),
	(3, 1, 1, 0, 2048, True, True, 0.0)])
def test_layernorm_backward(A, seq_in, seq_out, seq_off, E, has_dx, g_out, drop):
	from novic_amd import ops
	rows = A * seq_in
	g = torch.Generator().manual_seed(A * 7 + E + seq_in)
	x = (torch.randn(rows, E, generator=g) * 2 + 0.25).requires_grad_(True)
	gamma = torch.randn(E, generator=g).requires_grad_(True)
	dy = torch.randn(A * seq_out, E, generator=g).bfloat16()
	dx_in = torch.randn(rows, E, generator=g) if has_dx else None
	y = _ln(x, gamma).view(A, seq_in, E)[:, seq_off:seq_off + seq_out].reshape(A * seq_out, E)
	y.backward(dy.float())
	want_dx = x.grad + (dx_in if has_dx else 0)
	dgamma0 = torch.randn(E, generator=g)

	dxo = torch.full((rows, E), float("nan"), device="cuda")
	gb = torch.empty(rows, E, dtype=torch.bfloat16, device="cuda") if g_out else None
	dgam = dgamma0.cuda()
	dr = ops.Dropout(drop, 99, 3)
	ops.layernorm_bwd(dy.cuda(), x.detach().cuda(), gamma.detach().cuda(), dx_in.cuda() if has_dx else None, dxo, gb, dgam, rows, E, seq_in=seq_in, seq_out=seq_out,
	                  seq_off=seq_off, dropout=dr)
	scale = float(want_dx.abs().max())
	assert float((dxo.cpu() - want_dx).abs().max()) <= 2e-5 * scale
	gs = float(gamma.grad.abs().max()) + 1.0
	assert float((dgam.cpu() - dgamma0 - gamma.grad).abs().max()) <= 1e-4 * gs   # fp32 atomics in arbitrary order over up to 1024 blocks
	if g_out:
		got = gb.float().cpu()
		if drop == 0.0:
			assert float((got - want_dx).abs().max()) <= scale * 2 ** -8
		else:  # every element is either dropped or scaled by 1/(1-p); the kept fraction matches p
			kept = got != 0
			assert float((got - want_dx / (1 - drop))[kept].abs().max()) <= scale / (1 - drop) * 2 ** -8
			frac = float(kept.float().mean())
			assert abs(frac - (1 - drop)) < 0.01
			# same site, same seed => the same mask again (the forward of the layer below draws it with identical arguments)
			gb2 = torch.empty_like(gb)
			ops.layernorm_bwd(dy.cuda(), x.detach().cuda(), gamma.detach().cuda(), dx_in.cuda() if has_dx else None, dxo, gb2, None, rows, E, seq_in=seq_in,
			                  seq_out=seq_out, seq_off=seq_off, dropout=dr)
			assert torch.equal(gb2 != 0, gb != 0)


def test_layernorm_backward_in_place():
	"""dx_out may alias dx_in (the training step accumulates the residual-stream gradient in one buffer)."""
	from novic_amd import ops
	rows, E = 5000, 512
	g = torch.Generator().manual_seed(5)
	x, gamma, dx = torch.randn(rows, E, generator=g).cuda(), torch.randn(E, generator=g).cuda(), torch.randn(rows, E, generator=g).cuda()
	dy = torch.randn(rows, E, generator=g).bfloat16().cuda()
	out = torch.empty_like(dx)
	ops.layernorm_bwd(dy, x, gamma, dx, out, None, None, rows, E)
	ops.layernorm_bwd(dy, x, gamma, dx, dx, None, None, rows, E)
	assert torch.equal(out, dx)
