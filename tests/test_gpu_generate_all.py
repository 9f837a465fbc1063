"""generate_all on the GPU (reference embedding_decoder.py:986-1079): exactness of the two kernels on given inputs, then the whole call against the
reference-generated fixtures (scores within bf16 tolerance; the selected targets may differ where two scores are closer than that)."""
import pytest
import torch

from conftest import load_golden
from helpers import make_decoder
from oracle import decoder_oracle as O

pytestmark = pytest.mark.gpu
ALL = load_golden("decoder_generate_all.pt")


@pytest.mark.parametrize("B,W,K,ties", [(5, 1000, 10, False), (3, 42919, 10, False), (4, 300, 32, True), (2, 40, 40, True)])
def test_topk_rows_exact(B, W, K, ties):
	from novic_amd import ops
	g = torch.Generator().manual_seed(W)
	s = torch.randn(B, W, generator=g)
	if ties:
		s = (s * 2).round() / 2
	adj, sc = torch.rand(W, generator=g), 0.5 + torch.rand(W, generator=g)
	val = torch.empty(B, K).cuda()
	idx = torch.empty(B, K, dtype=torch.int32).cuda()
	ops.topk_rows(s.cuda(), K, val, idx, adjust=adj.cuda(), adjust_scale=0.7, scale=sc.cuda())
	eff = (s - 0.7 * adj) * sc
	order = torch.sort(eff, dim=1, descending=True, stable=True).indices[:, :K]  # stable = ties towards the lower index
	assert torch.equal(idx.cpu().long(), order)
	torch.testing.assert_close(val.cpu(), eff.gather(1, order), atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("renorm", [False, True])
def test_score_targets_against_torch(renorm):
	from novic_amd import ops
	from novic_amd.guide_trie import TokenTrie
	B, Hc, T, V, tau = 3, 7, 5, 307, 1.7
	g = torch.Generator().manual_seed(5)
	rows = set()
	while len(rows) < Hc:
		ln = int(torch.randint(1, T, (1,), generator=g))
		rows.add(tuple([int(torch.randint(1, 4, (1,), generator=g))] + [int(t) for t in torch.randint(1, 6, (ln - 1,), generator=g)]))
	tg = torch.zeros(Hc, T, dtype=torch.int64)
	for i, r in enumerate(sorted(rows)):
		tg[i, :len(r)] = torch.tensor(r)
	trie = TokenTrie(tg, torch.device("cuda"))
	valid = torch.from_numpy(trie.path_node_host >= 0)
	Vp = (V + 7) // 8 * 8
	logits = torch.randn(B * Hc * T, Vp, generator=g).to(torch.bfloat16)
	out = torch.full((B, Hc + 3), float("nan")).cuda()
	node = torch.from_numpy(trie.path_node_host.clip(min=0)).cuda() if renorm else None
	ops.score_targets(logits.cuda(), Vp, V, tg.cuda(), (~valid).to(torch.uint8).cuda(), node, trie, out, 2, B, Hc, T, tau)
	lg = logits[:, :V].float().view(B, Hc, T, V) / tau
	ref = torch.zeros(B, Hc)
	for h in range(Hc):
		consistent = torch.ones(Hc, dtype=torch.bool)
		for t in range(T):
			if not valid[h, t]:
				break
			x = lg[:, h, t].clone()
			if renorm:
				allowed = torch.zeros(V, dtype=torch.bool)
				allowed[tg[consistent, t]] = True
				x[:, ~allowed] = float("-inf")
			ref[:, h] += torch.log_softmax(x, dim=1)[:, tg[h, t]]
			consistent &= tg[:, t] == tg[h, t]
	torch.testing.assert_close(out.cpu()[:, 2:2 + Hc], ref, atol=2e-4, rtol=1e-5)
	assert torch.isnan(out.cpu()[:, :2]).all() and torch.isnan(out.cpu()[:, 2 + Hc:]).all()


@pytest.mark.parametrize("case", ALL, ids=[c["name"] for c in ALL])
def test_generate_all_against_reference_fixture(case):
	spec = O.DecoderSpec(**case["spec"])
	sd = O.init_state_dict(spec, seed=case["seed"])
	model, _ = make_decoder(spec, token_dtype=torch.int64, sd=sd, device="cuda")
	model.eval()
	guide = case["guide_targets"].cuda()
	v_arg = (case["vocab_targets"].cuda() if case.get("vocab_targets") is not None else guide) if case["vocab_prior"] else None
	args = dict(topk=case["topk"], temperature=case["temperature"], length_alpha=case["length_alpha"], vocab_targets=v_arg, vocab_per_token=case["vocab_per_token"],
	            vocab_scaler=case["vocab_scaler"], guide_targets=guide, guide_renorm=case["guide_renorm"])
	with torch.no_grad():
		ids, pad, score = model.generate_all(embed=case["embed"].cuda(), **args)
		pre = model.precompute_generate_all(**{k: v for k, v in args.items() if k not in ("topk", "temperature")})
		ids2, pad2, score2 = model.generate_all(embed=case["embed"].cuda(), precompute=pre, **args)
	assert torch.equal(ids, ids2) and torch.equal(pad, pad2) and torch.equal(score, score2)
	ids, pad, score = ids.cpu(), pad.cpu(), score.cpu()
	assert ids.shape == case["ids"].shape and pad.dtype == torch.bool
	fin = torch.isfinite(case["score"])
	assert torch.equal(fin, torch.isfinite(score))   # guide targets the vocabulary does not contain score -inf
	assert torch.all(score[:, :-1] >= score[:, 1:])
	# every returned row is one of the guide targets, none twice
	gset = {tuple(r.tolist()) for r in case["guide_targets"][:, :ids.shape[2]]}
	for b in range(ids.shape[0]):
		rows = [tuple(r.tolist()) for r in ids[b]]
		assert len(set(rows)) == len(rows) and all(r in gset for r in rows)
	# scores: the fixture's best within bf16 tolerance, and where the same target was selected the scores agree
	assert float((score[:, 0] - case["score"][:, 0]).abs().max()) <= 6e-2
	same = (ids == case["ids"]).all(dim=2) & fin
	assert same.float().sum().item() >= 0.7 * fin.float().sum().item()
	torch.testing.assert_close(score[same], case["score"][same], atol=6e-2, rtol=1e-2)
