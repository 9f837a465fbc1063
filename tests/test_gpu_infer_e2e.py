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


def test_checkpoint_to_labels(tmp_path):
	from novic_amd import clip_vit, embedders, embedding_dataset, embedding_decoder, infer, train, utils
	spec_path = tmp_path / "embedder.json"
	spec_path.write_text(json.dumps(dict(tokens=TOKENS, embed_dim=64)))
	emb = embedders.Embedder.create(f"local:{spec_path}", device="cuda")
	tc = emb.create_target_config(NOUNS, **embedding_decoder.PrefixedIterDecoder.get_target_config_kwargs(
		with_start_token=False, with_end_token=True, compact_ids=True, fixed_token_length=False, auto_fixed_token_length=True, use_masks=True))
	emb.configure_target(tc, NOUNS)
	cfg_flat = _cfg_flat(f"local:{spec_path}")
	torch.manual_seed(0)
	model = infer.load_decoder_model(utils.AttrDict.from_dict(cfg_flat), emb, embedding_dataset.DataConfig.single(), None).cuda()
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
		nm.set_gencfg("greedy_k1_vnone_gn_t1_a0")
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
