#!/usr/bin/env python3
"""The decoder's non-default constructor switches at the BENCH layer size (6 layers, d = 512, V = 6912; micro-batches of 512 x 4 per optimizer step): a dozen optimizer steps with
dropout, noise-free, per combination -- the loss must fall and every parameter stay finite -- and the time of a step beside the released recipe's (the switches run on the
general kernels: correct, not tuned).  python tools/variant_scale_check.py"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from novic_amd import embedding_dataset, embedding_decoder, train as T  # noqa: E402

dev = torch.device("cuda")
spec = bench.WorkloadSpec(embed_dim=bench.F_DIM, vocab_size=bench.VOCAB, token_length=bench.CMAX)


def build(**kw):
	dc = embedding_dataset.DataConfig.create(dict(use_weights=False, unit_weights=True, multi_target=False, multi_first=False, full_targets=True, fixed_multi_length=True, multi_length=1))
	cfg = dict(vocab_quant=False, num_end_loss=1, label_smoothing=0.0, hidden_dim=spec.hidden_dim, feedfwd_scale=f"{spec.feedfwd_dim}/{spec.hidden_dim}", mlp_hidden_layer="none",
	           mlp_hidden_bias=False, mlp_hidden_norm=False, mlp_hidden_activation="gelu", input_dropout=0.1, num_layers=spec.num_layers, num_heads=spec.num_heads, layer_dropout=0.1,
	           layer_activation="gelu", layer_norm_first=True, layer_bias=False, logits_bias=False, init_bias_zero=True, init_mlp_mode="balanced", init_mlp_unit_norm=False,
	           init_tfrm_mode="balanced", init_tfrm_unit_norm=False, init_tfrm_unit_postnorm=True, init_tfrm_proj_layers=True, init_zero_norm=False, init_rezero_mode="none",
	           mlp_seq_len=spec.mlp_seq_len, weight_tying=True, strictly_causal=False, enable_nested=False)
	cfg.update(kw)
	return embedding_decoder.PrefixedIterDecoder(embedder=bench._CachedTextEmbedder(spec), data_config=dc, **cfg).to(dev)


for name, kw in (("released recipe", {}),
                 ("layer_bias + relu", dict(layer_bias=True, layer_activation="relu", init_bias_zero=False)),
                 ("post-LN + ReZero perskip + every bias + MLP gmean / norm + untied", dict(layer_norm_first=False, init_rezero_mode="perskip", layer_bias=True, mlp_hidden_layer="gmean",
                                                                                           mlp_hidden_norm=True, mlp_hidden_bias=True, logits_bias=True, weight_tying=False)),
                 ("pre-LN ReZero perlayer + tanh", dict(init_rezero_mode="perlayer", layer_activation="tanh"))):
	torch.manual_seed(0)
	model = build(**kw)
	model.train()
	opt = T.FusedAdamW(model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.1, max_norm=1.0)
	mbs = [bench.synth_micro_batch(spec, 512, seed=100 + i, device=dev) for i in range(4)]
	losses = []
	for step in range(12):
		stats, gnorm = T.train_step(model, opt, mbs)
		if step in (0, 3, 7, 11):
			torch.cuda.synchronize()
			losses.append(round(float(stats[1].sum() / stats[0].sum()), 4))
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(10):
		T.train_step(model, opt, mbs)
	torch.cuda.synchronize()
	ms = (time.perf_counter() - t0) * 100
	ok = bool(torch.isfinite(model.flat_parameters()).all()) and losses[-1] < losses[0]
	print(f"{name}: loss at steps 1 / 4 / 8 / 12 {losses}, parameters finite and loss falling: {ok}, {ms:.2f} ms per 2048-sample step", flush=True)
	assert ok, name
