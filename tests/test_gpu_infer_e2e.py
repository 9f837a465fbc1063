"""End-to-end plumbing on the GPU (config 1 of BASELINE.json with a synthetic checkpoint, since released checkpoints / CLIP weights are not
reachable offline): train-loop checkpoint -> infer.NOVICModel -> image tensor -> native ViT embedding -> beam / greedy labels as strings."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

TOKENS = ["red", "panda", "fire", "truck", "sea", "lion", "ice", "cream", "cone", "dog", "cat", "house"]
NOUNS = ("red panda", "fire truck", "sea lion", "ice cream cone", "dog", "cat", "house", "dog house", "sea")


def _cfg_flat(embedder_spec):
	return dict(model="PrefixedIterDecoder", embedder_spec=embedder_spec, embedder_amp=True, embedder_amp_bf16=True, embedder_compile=False, embedder_optimum=False, amp=True,
	            amp_bf16=True, vocab_quant=False, num_end_loss=1, label_smoothing=0.0, hidden_dim=64, feedfwd_scale="1/4", mlp_hidden_layer="none", mlp_hidden_bias=False,
	            mlp_hidden_norm=False, mlp_hidden_activation="gelu", input_dropout=0.1, num_layers=2, num_heads=4, layer_dropout=0.1, layer_activation="gelu",
	            layer_norm_first=True, layer_bias=False, logits_bias=False, init_bias_zero=True, init_mlp_mode="balanced", init_mlp_unit_norm=False, init_tfrm_mode="balanced",
	            init_tfrm_unit_norm=False, init_tfrm_unit_postnorm=True, init_tfrm_proj_layers=True, init_zero_norm=False, init_rezero_mode="none", mlp_seq_len=4,
	            weight_tying=True, strictly_causal=False, enable_nested=False)


EVERY_SWITCH = dict(layer_norm_first=False, layer_bias=True, init_rezero_mode="perskip", layer_activation="relu", mlp_hidden_layer="gmean", mlp_hidden_bias=True, mlp_hidden_norm=True,
                    mlp_hidden_activation="tanh", logits_bias=True, weight_tying=False, init_bias_zero=False)


@pytest.mark.parametrize("switches", [{}, EVERY_SWITCH], ids=["released_recipe", "every_switch"])
def test_checkpoint_to_labels(tmp_path, switches):
	"""every_switch: the same journey with every non-default switch of the reference's decoder constructor on at once (post-LN layers with ReZero from zero-initialised scalars,
	layer / logits / MLP biases, relu, a normalised tanh hidden layer in the prefix MLP, an untied token table; round 5) -- the general kernels under dropout LEARN the nine
	pairs, and the checkpoint round-trips through the reference's `.model` format and key names."""
	from novic_amd import clip_vit, embedders, embedding_dataset, embedding_decoder, infer, train, utils
	spec_path = tmp_path / "embedder.json"
	spec_path.write_text(json.dumps(dict(tokens=TOKENS, embed_dim=64)))
	emb = embedders.Embedder.create(f"local:{spec_path}", device="cuda")
	tc = emb.create_target_config(NOUNS, **embedding_decoder.PrefixedIterDecoder.get_target_config_kwargs(
		with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True))
	emb.configure_target(tc, NOUNS)
	cfg_flat = dict(_cfg_flat(f"local:{spec_path}"), **switches)
	torch.manual_seed(0)
	model = infer.load_decoder_model(utils.AttrDict.from_dict(cfg_flat), emb, embedding_dataset.DataConfig.single(), None).cuda()
	assert model._general_layers == bool(switches)
	# a few real optimizer steps on (random embedding -> noun) pairs so the checkpoint is a trained-loop artefact
	opt = train.FusedAdamW(model, lr=3e-3)
	model.train()
	ids, mask = emb.tokenize_target(NOUNS)
	g = torch.Generator().manual_seed(1)
	proto = torch.nn.functional.normalize(torch.randn(len(NOUNS), 64, generator=g), dim=-1)
	for _ in range(400):  # memorise 9 (embedding -> noun) pairs: also an end-to-end check that HIP forward/backward/AdamW actually learn
		train.train_step(model, opt, [(proto.clone().cuda(), ids.cuda(), mask.cuda(), None)])
	path = train.save_train_checkpoint(cfg_flat, model, None, None, ("",) + NOUNS, 1, None, None, model_only=True, run_dir=str(tmp_path), chunk_id=3)
	assert path.endswith(".model") and os.path.exists(path)

	vit = clip_vit.NativeViT(clip_vit.ViTConfig(image_size=64, patch_size=16, width=128, layers=2, heads=4, embed_dim=64), seed=2).cuda()
	emb2 = embedders.Embedder.create(f"local:{spec_path}", device="cuda")
	emb2.attach_image_tower(vit)
	nm = infer.NOVICModel(path, gencfg="beam_k3_vnone_gn_t1_a0", batch_size=8, device="cuda", embedder=emb2)
	images = torch.randn(5, 3, 64, 64, generator=g)
	with nm:
		out = nm.classify_images(images)
		embeds = nm.embed_images(images)
		direct = nm.decoder.generate_beam(embed=embeds, topk=3, temperature=1.0, length_alpha=0.0, vocab_targets=None, vocab_per_token=False, vocab_scaler=0.0,
		                                  guide_targets=None, guide_renorm=False)
		assert nm.is_decoder_loaded()
		# pipelined batches (the tower of the next batch beside the decoding of the current one, its GEMM grids on fewer CUs): the predictions of one call per batch
		more = [torch.randn(n, 3, 64, 64, generator=g) for n in (5, 5, 3)]
		one_by_one = [nm.classify_images(b) for b in more]
		piped = list(nm.classify_image_batches(more))
		assert len(piped) == 3 and not torch.is_inference_mode_enabled()
		for a, b in zip(one_by_one, piped):
			assert a.preds == b.preds and a.types == b.types and torch.equal(a.embeds, b.embeds) and a.logprobs == b.logprobs
		assert list(nm.classify_image_batches([])) == []
		# (round 6) two decode calls at the same time on lanes of their own (decode_lanes = 2, the default): seven batches of one shape, one batch per decode call
		# (decode_rows = 1), so that pairs form and one call is left over -- against one call at a time, the latency mode and one call per batch
		same = [torch.randn(4, 3, 64, 64, generator=g) for _ in range(7)]
		ref = [nm.classify_images(b) for b in same]
		same_dev = [b.cuda() for b in same]  # (lanes are for batches that are already on the device; host batches keep one decode call at a time)
		from novic_amd import embedding_decoder as ED
		many_calls = []
		orig_many = ED.PrefixedIterDecoder.generate_beam_many
		ED.PrefixedIterDecoder.generate_beam_many = lambda self, embeds, *a: (many_calls.append(len(embeds)), orig_many(self, embeds, *a))[1]
		try:
			for src, kw, lanes_seen in ((same_dev, dict(decode_rows=1), [2, 2, 2]), (same_dev, dict(decode_rows=1, decode_lanes=1), []), (same_dev, dict(decode_rows=1, decode_lanes=3), [3, 3]),
			                            (same_dev, dict(latency=True), []), (same_dev + same_dev[:1], dict(), [2]), (same_dev, dict(), []), (same, dict(decode_rows=1), []), (same, dict(), [])):  # (defaults: four batches per tower launch and decode call -- eight batches make a pair of equal calls, seven a call of 16 rows and one of 12: one after the other)
				many_calls.clear()
				got = list(nm.classify_image_batches(src, **kw))
				assert len(got) == len(src), kw
				assert many_calls == lanes_seen, (kw, src[0].device, many_calls)
				for a, b in zip(ref + ref[:1], got):
					assert a.preds == b.preds and a.types == b.types and torch.equal(a.embeds, b.embeds) and a.logprobs == b.logprobs, kw
		finally:
			ED.PrefixedIterDecoder.generate_beam_many = orig_many
		nm.set_gencfg("greedy_k1_vnone_gn_t1_a0")
		for kw in (dict(decode_rows=1), dict(decode_rows=1, decode_lanes=1)):
			gref = [nm.classify_images(b) for b in same[:5]]
			got = list(nm.classify_image_batches(same_dev[:5], **kw))
			for a, b in zip(gref, got):
				assert a.preds == b.preds and a.types == b.types and torch.equal(a.embeds, b.embeds) and a.logprobs == b.logprobs, kw
		greedy = nm.classify_embeds(embeds)
		# the decoder learnt the prototypes: decoding a prototype embedding returns its noun
		proto_out = nm.classify_embeds(proto.cuda())
	assert not nm.is_decoder_loaded()
	assert out.embeds.shape == (5, 64) and len(out.preds) == 5 and all(len(p) == 3 for p in out.preds)
	assert all(isinstance(s, str) for p in out.preds for s in p)
	for lp, pr in zip(out.logprobs, out.probs):
		assert all(a >= b for a, b in zip(lp, lp[1:])) and all(0 < p <= 1 and abs(p - math.exp(l)) < 1e-9 for p, l in zip(pr, lp))
	assert emb2.detokenize_target(direct[0].cpu()) == [list(p) for p in out.preds]
	assert all(len(p) == 1 for p in greedy.preds)
	assert all(t in (infer.PredictionType.ValidGuide, infer.PredictionType.ValidVocab, infer.PredictionType.Other) for row in out.types for t in row)
	hits = sum(p[0] == n for p, n in zip(proto_out.preds, NOUNS))
	assert hits >= len(NOUNS) - 1, (hits, proto_out.preds)
	# the default generation config is guided beam search over the model's own nouns (infer.py:275): every live beam is one of the nouns
	nm2 = infer.NOVICModel(path, device="cuda", embedder=emb2)
	assert nm2.gencfg.name == "beam_k10_vnone_gp_t1_a0"
	with nm2:
		guided = nm2.classify_embeds(proto.cuda())
		nm2.set_gencfg("greedy_k1_vnone_gr_t1_a0")
		guided_greedy = nm2.classify_embeds(proto.cuda())
		nm2.set_gencfg("all_k3_vtgt1_gp_t1_a0")   # every noun scored by teacher forcing (generate_all), vocabulary prior on
		scored_all = nm2.classify_embeds(proto.cuda())
	for preds, lps, types in zip(guided.preds, guided.logprobs, guided.types):
		live = [p for p, l in zip(preds, lps) if math.isfinite(l)]
		assert len(live) == len(NOUNS) and set(live) == set(NOUNS)      # 9 nouns < 10 beams: the tail is dead (-inf), the rest enumerate the noun set
		assert all(t == infer.PredictionType.ValidGuide for t, l in zip(types, lps) if math.isfinite(l))
	assert sum(p[0] == n for p, n in zip(guided.preds, NOUNS)) >= len(NOUNS) - 1
	assert all(p[0] in NOUNS for p in guided_greedy.preds) and sum(p[0] == n for p, n in zip(guided_greedy.preds, NOUNS)) >= len(NOUNS) - 1
	assert all(len(p) == 3 and len(set(p)) == 3 and set(p) <= set(NOUNS) for p in scored_all.preds)
	assert sum(p[0] == n for p, n in zip(scored_all.preds, NOUNS)) >= len(NOUNS) - 1


def test_evaluation_callers(tmp_path):
	"""eval_top1 / GenerationTaskList / eval_cls_decoding / infer_predictions (reference train.py:170-240, :1726-1868, :2337-2450, :2606-2724) on a decoder that has
	memorised nine (embedding -> noun) pairs: the aggregates must be what the per-batch decoder outputs imply."""
	from novic_amd import embedders, embedding_dataset, embedding_decoder, evaluate, infer, train, utils
	spec_path = tmp_path / "embedder.json"
	spec_path.write_text(json.dumps(dict(tokens=TOKENS, embed_dim=64)))
	emb = embedders.Embedder.create(f"local:{spec_path}", device="cuda")
	tc = emb.create_target_config(NOUNS, **embedding_decoder.PrefixedIterDecoder.get_target_config_kwargs(
		with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True))
	emb.configure_target(tc, NOUNS)
	torch.manual_seed(0)
	dc = embedding_dataset.DataConfig.single()
	model = infer.load_decoder_model(utils.AttrDict.from_dict(_cfg_flat(f"local:{spec_path}")), emb, dc, None).cuda()
	opt = train.FusedAdamW(model, lr=3e-3)
	model.train()
	ids, mask = emb.tokenize_target(NOUNS)
	proto = torch.nn.functional.normalize(torch.randn(len(NOUNS), 64, generator=torch.Generator().manual_seed(1)), dim=-1)
	for _ in range(400):
		train.train_step(model, opt, [(proto.clone().cuda(), ids.cuda(), mask.cuda(), None)])
	model.eval()
	guide = infer.load_guide_targets(NOUNS, emb, torch.device("cuda"), False)

	# ---- eval_top1: two batches (5 + 4 samples); one target corrupted so that it is wrong ----
	bad = ids.clone()
	bad[0, 0] = ids[1, 0]
	batches = [(proto[:5].cuda(), bad[:5].cuda(), mask[:5].cuda(), None), (proto[5:].cuda(), bad[5:].cuda(), mask[5:].cuda(), None)]
	for guided in (None, guide):
		loss, noun_top1, top1, top1_seq, n_tok, n_valid, n_samples, n_batches, _ = evaluate.eval_top1(model, batches, dc, tc.token_length, guide_token_ids=guided)
		assert (n_samples, n_batches, n_valid) == (9, 2, 9) and n_tok == int((~mask).sum())
		# the corrupted first token is wrong, and so may be the token predicted right after it (teacher forcing feeds the wrong token)
		assert noun_top1 == pytest.approx(8 / 9) and any(top1 == pytest.approx((n_tok - k) / n_tok) for k in (1, 2)) and math.isfinite(loss) and loss > 0
		assert len(top1_seq) == tc.token_length and top1_seq[0] == pytest.approx(8 / 9) and top1_seq[1] >= 8 / 9 - 1e-6

	# ---- generation task list over several decoding strategies, class lists = the nouns themselves (+ an alias for class 0) ----
	gencfgs = [infer.GenerationConfig.from_name(n) for n in ("greedy_k1_vnone_gn_t1_a0", "beam_k3_vnone_gp_t1_a0", "all_k3_vtgt1_gp_t1_a0")]
	class_lists = [(n,) for n in NOUNS]
	tl = evaluate.GenerationTaskList(gencfgs, model, set(NOUNS), guide, set(NOUNS), guide, class_lists=class_lists)
	assert len(tl) == 3 and tl[1].gencfg.topk == 3
	res = evaluate.eval_cls_decoding(tl, [(proto[:4], list(range(4)), None), (proto[4:], list(range(4, 9)), None)], torch.device("cuda"))
	for (gcfg, topk, topk_guide, topk_vocab, topk_invalid), task in zip(res, tl):
		assert gcfg is task.gencfg and topk.shape == (gcfg.topk,) and task.num_samples == 9
		assert float(topk[0]) >= 8 / 9 - 1e-6 and float(topk_invalid[0]) <= 1 / 9 + 1e-6
		assert torch.all(topk[1:] >= topk[:-1]) and torch.all(topk_guide >= topk)      # top-k ratios are cumulative; correct implies valid
	# the same statistics when equal-shaped batches are decoded concurrently (two lanes) and one at a time
	batches = [(proto[:4], list(range(4)), None), (proto[4:8], list(range(4, 8)), None), (proto[8:], [8], None)]
	one = [(float(t[1][0]), float(t[4][0])) for t in evaluate.eval_cls_decoding(tl, batches, torch.device("cuda"), lanes=1)]
	strs_one = [list(task.target_str) for task in tl]
	two = [(float(t[1][0]), float(t[4][0])) for t in evaluate.eval_cls_decoding(tl, batches, torch.device("cuda"), lanes=2)]
	assert one == two and [list(task.target_str) for task in tl] == strs_one and all(task.num_samples == 9 for task in tl)
	# generate_many hands every batch's outputs to on_batch, in order, once every task has seen every batch
	seen = []
	with torch.inference_mode():
		outs = tl.generate_many([proto[:4].cuda(), proto[4:8].cuda()], [list(range(4)), list(range(4, 8))], on_batch=lambda b, per_task: seen.append((b, per_task)))
	assert [b for b, _ in seen] == [0, 1] and all(len(per_task) == len(tl) for _, per_task in seen)
	assert all(seen[b][1][t] is outs[t][b] for b in range(2) for t in range(len(tl)))
	# ---- infer_predictions + the predictions JSON ----
	preds = evaluate.infer_predictions(tl, [(["a", "b", "c"], proto[:3].cuda()), (["d"], proto[3:4].cuda())])
	assert set(preds) == {g.name for g in gencfgs} and list(preds["beam_k3_vnone_gp_t1_a0"]) == ["a", "b", "c", "d"]
	assert [p[0][0] for p in preds["greedy_k1_vnone_gn_t1_a0"].values()] == list(NOUNS[:4])
	assert all(len(v) == 3 and v[0][1] >= v[1][1] >= v[2][1] for v in preds["all_k3_vtgt1_gp_t1_a0"].values())
	out = evaluate.write_pred_json(str(tmp_path / "pred.json"), tl, preds, model_path="tiny.model", guide_targets=NOUNS, vocab_targets=NOUNS)
	doc = json.load(open(out))
	assert doc["version"] == 1 and doc["samples"] == ["a", "b", "c", "d"] and doc["predictions"]["beam_k3_vnone_gp_t1_a0"]["pred"][0][0] == NOUNS[0]
