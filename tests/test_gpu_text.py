"""GPU parity of the native CLIP text tower against the fixtures generated from transformers.CLIPTextModelWithProjection (tests/golden/
make_golden_text.py) and against the CPU oracle with the kernels' bf16 rounding points.  Tolerance (stated): per-row L2 error of the unit embedding
<= 3e-2 against the fp32 fixture (bf16 GEMM operands over up to 2 layers here; cosine >= 0.999) and <= 1.5e-2 against the bf16-emulating oracle."""
import pytest
import torch

from conftest import load_golden
from oracle import text_oracle as TO

pytestmark = pytest.mark.gpu
CASES = load_golden("text_forward.pt")


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_text_tower_matches_fixture(case):
	from novic_amd import clip_text
	spec = TO.TextSpec(**case["spec"])
	sd = TO.init_state_dict(spec, seed=case["seed"])
	tower = clip_text.NativeTextTower(clip_text.TextConfig(**case["spec"]))
	tower.load_state_dict(sd)
	tower.cuda()
	ids = case["token_ids"]
	out = tower(ids.cuda()).cpu()
	out32 = tower(ids.to(torch.int32).cuda()).cpu()
	assert torch.equal(out, out32)
	assert out.shape == case["embeds"].shape and torch.isfinite(out).all()
	torch.testing.assert_close(out.norm(dim=1), torch.ones(out.shape[0]), atol=1e-5, rtol=0)
	err = (out - case["embeds"]).norm(dim=1)
	assert float(err.max()) <= 3e-2, err
	emu = TO.encode_text(sd, spec, ids, bf16=True)
	assert float((out - emu).norm(dim=1).max()) <= 1.5e-2
	# causality: tokens after the END-OF-TEXT token do not change its embedding (so the reference's padding to the context length need not be computed)
	lens = ids.argmax(dim=1) + 1
	L = int(lens.max())
	if L < ids.shape[1]:
		torch.testing.assert_close(tower(ids[:, :L].cuda()).cpu(), out, atol=2e-3, rtol=0)


def test_text_tower_hf_names_and_embedder_hook():
	from novic_amd import clip_text, embedders
	case = CASES[0]
	spec = TO.TextSpec(**case["spec"])
	sd = TO.init_state_dict(spec, seed=case["seed"])
	W = spec.width
	hf = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"], "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
	      "text_model.final_layer_norm.weight": sd["ln_final.weight"], "text_model.final_layer_norm.bias": sd["ln_final.bias"], "text_projection.weight": sd["text_projection"].T.contiguous()}
	for i in range(spec.layers):
		o, h = f"transformer.resblocks.{i}.", f"text_model.encoder.layers.{i}."
		for j, nm in enumerate(("q_proj", "k_proj", "v_proj")):
			hf[h + f"self_attn.{nm}.weight"], hf[h + f"self_attn.{nm}.bias"] = sd[o + "attn.in_proj_weight"][j * W:(j + 1) * W], sd[o + "attn.in_proj_bias"][j * W:(j + 1) * W]
		for a, b in (("self_attn.out_proj", "attn.out_proj"), ("layer_norm1", "ln_1"), ("layer_norm2", "ln_2"), ("mlp.fc1", "mlp.c_fc"), ("mlp.fc2", "mlp.c_proj")):
			hf[h + a + ".weight"], hf[h + a + ".bias"] = sd[o + b + ".weight"], sd[o + b + ".bias"]
	tower = clip_text.NativeTextTower(clip_text.TextConfig(**case["spec"]))
	tower.load_hf_state_dict(hf)
	tower.cuda()
	ref = case["embeds"]
	assert float((tower(case["token_ids"].cuda()).cpu() - ref).norm(dim=1).max()) <= 3e-2
	# through the embedder surface: inference_text = tokenize + inference_tokens (reference embedders.py:423-426); pooling at the embedder's end token
	toks = [f"w{i}" for i in range(40)]
	emb = embedders.LocalVocabEmbedder(toks, embed_dim=spec.embed_dim, context_length=spec.context_length, device="cuda")
	t2 = clip_text.NativeTextTower(clip_text.TextConfig(**{**case["spec"], "vocab_size": 64}), seed=5, eot_token_id=emb.end_token_id).cuda()
	emb.attach_text_tower(t2)
	with emb.inference_mode():
		e1 = emb.inference_text(["w1 w2 w3", "w7"])
		e2 = emb.inference_tokens(emb.tokenize(["w1 w2 w3", "w7"], output_dict=True))
	assert e1.shape == (2, spec.embed_dim) and torch.equal(e1, e2)
	torch.testing.assert_close(e1.norm(dim=1).cpu(), torch.ones(2), atol=1e-5, rtol=0)
	sd2 = {k: v.cpu() for k, v in t2.state_dict().items()}
	ids = emb.tokenize(["w1 w2 w3", "w7"], output_dict=True)["input_ids"]
	emu = TO.encode_text(sd2, TO.TextSpec(**{**case["spec"], "vocab_size": 64}), ids, bf16=True, eot_token_id=emb.end_token_id)
	assert float((e1.cpu() - emu).norm(dim=1).max()) <= 1.5e-2


def test_text_tower_at_bench_size_against_the_oracle():
	"""The text tower bench.py measures (ViT-B/32's text side: 12 layers, width 512, 8 heads, 77-token rows, QuickGELU, vocabulary 49 408) at ITS batch, 256 texts =
	19 712 token rows, so that the 256-wide tiles run (asserted through the tile counters): EVERY one of the 256 pooled embeddings against the oracle tower in fp32 (round 3
	sampled 24 rows), 64 of them also against its bf16 emulation of the kernels' rounding points."""
	from novic_amd import clip_text, ops
	dims = dict(vocab_size=49408, context_length=77, width=512, layers=12, heads=8, mlp_ratio=4.0, embed_dim=512, quick_gelu=True)
	spec = TO.TextSpec(**dims)
	sd = TO.init_state_dict(spec, seed=31)
	tower = clip_text.NativeTextTower(clip_text.TextConfig(**dims))
	tower.load_state_dict(sd)
	tower.cuda()
	g = torch.Generator().manual_seed(32)
	ids = torch.randint(1, 49406, (256, 77), generator=g)
	lens = torch.randint(4, 77, (256,), generator=g)
	ids[:, 0] = 49406
	for b in range(256):
		ids[b, int(lens[b])] = 49407          # END-OF-TEXT (the largest id: the pooling position)
		ids[b, int(lens[b]) + 1:] = 0
	ops.gemm_tile_counts(reset=True)
	out = tower(ids.cuda()).cpu()
	counts = ops.gemm_tile_counts()
	assert counts["t256"] >= 4 * 12 and counts["skinny"] == 0, counts  # QKV, out-projection (late round 4: mid-size host-row-count projections left the streaming kernel), fc1, fc2 of every layer on the 256-wide tile
	ref = torch.cat([TO.encode_text(sd, spec, ids[i:i + 64]) for i in range(0, 256, 64)])
	assert out.shape == ref.shape == (256, 512)
	assert float((out * ref).sum(dim=1).min()) >= 0.999 and float((out - ref).norm(dim=1).max()) <= 3e-2
	pick = torch.arange(0, 256, 4)
	emu = TO.encode_text(sd, spec, ids[pick], bf16=True)
	assert float((out[pick] - emu).norm(dim=1).max()) <= 1.5e-2
	# round 6: the same tower with its residual stream in IEEE half -- clip's fp16 text tower, which the reference runs for 'openai:' embedders (embedders.py:488-489;
	# local_clip.OpenAIEmbedder switches it on) -- on the 256-wide tiles with the RESID_F16 epilogue: the same gates against the fp32 oracle, and close to (never equal to)
	# the fp32-stream tower
	tower.half_stream = True
	ops.gemm_tile_counts(reset=True)
	half = tower(ids.cuda()).cpu()
	assert torch.equal(tower(ids.cuda()).cpu(), half)  # (graph replay)
	counts = ops.gemm_tile_counts()
	assert counts["t256"] >= 4 * 12 and counts["skinny"] == 0 and counts["t192"] == 0, counts
	assert float((half * ref).sum(dim=1).min()) >= 0.999 and float((half - ref).norm(dim=1).max()) <= 3e-2
	assert float((half - out).norm(dim=1).max()) <= 1.5e-2 and not torch.equal(half, out)


@pytest.mark.parametrize("name", ["tiny_quick_short", "b32_depth2_ctx77", "b32_full_ctx77"])
def test_half_precision_text_tower_against_transformers_in_float16(name):
	"""Round 6: `NativeTextTower.half_stream` -- the residual stream as IEEE half, what the reference runs for 'openai:' embedders (clip's fp16 model) -- against transformers'
	text tower run in torch.float16 (tests/golden/text_forward_half.pt: full depth at ViT-B/32's text dimensions among the cases), against the oracle's restatement of clip's
	half-precision tower, and at the tight gate against the oracle with exactly this tower's rounding points (bf16 GEMM operands, half stream)."""
	from novic_amd import clip_text
	case = next(c for c in load_golden("text_forward_half.pt") if c["name"] == name)
	spec = TO.TextSpec(**case["spec"])
	sd = TO.init_state_dict(spec, case["seed"])
	tower = clip_text.NativeTextTower(clip_text.TextConfig(**case["spec"]))
	tower.load_state_dict(sd)
	tower.cuda()
	tower.half_stream = True
	ids = case["token_ids"]
	out = tower(ids.cuda()).cpu()
	assert torch.allclose(out.norm(dim=1), torch.ones(out.shape[0]), atol=1e-5)
	for ref in (case["embeds_half"], case["embeds_fp32"], TO.encode_text_half(sd, spec, ids)):
		assert float((out * ref).sum(dim=1).min()) >= 0.999 and float((out - ref).norm(dim=1).max()) <= 3e-2
	emu = TO.encode_text(sd, spec, ids, bf16=True, half_stream=True)
	assert float((out - emu).norm(dim=1).max()) <= 1.5e-2
	tower.half_stream = False
	full = tower(ids.cuda()).cpu()
	assert float((out - full).norm(dim=1).max()) <= 1.5e-2 and not torch.equal(out, full)
