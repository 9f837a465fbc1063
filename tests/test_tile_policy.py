"""The GEMM tile policy, pinned without a GPU (novic_gemm256_plan makes the decision of novic_gemm_bf16's 256-wide path without launching anything): for the shapes of
the training step, the towers at bench.py's batch and the decode steps -- which tile, how many workgroups, which K-split of the tiles behind the last whole round.
A rule put in front of a more specific one changes nothing in any result and only shows as time (round 3: the logits input gradient lost its device-planned tail that
way, 212 -> 278 us); this table is the guard."""
import pytest

from novic_amd import ops

R, B = ops.EPI_RESID_F32, ops.EPI_STORE_BF16
TABLE = [
	# name, (M, N, K), kwargs, (tile, workgroups, tail parts, tail tiles)
	("train qkv (allocated rows, device row count)", (81920, 1536, 512), dict(row_limit=True), (256, 256, 0, 0)),
	("train logits", (57344, 6912, 512), dict(row_limit=True), (256, 256, 0, 0)),
	("train logits dX: K-split tail planned on the device", (57344, 512, 6912), dict(row_limit=True, split_tail=True), (256, 256, -1, 0)),
	("train logits dX without scratch: no tail", (57344, 512, 6912), dict(row_limit=True), (256, 256, 0, 0)),
	("train in-proj dX", (81920, 512, 1536), dict(row_limit=True), (256, 256, 0, 0)),
	("train out-proj dX", (81920, 512, 512), dict(row_limit=True), (256, 256, 0, 0)),
	("train prefix MLP", (8192, 2048, 512), {}, (256, 256, 0, 0)),
	("ViT-B/32 qkv: two rounds need 232 workgroups", (12800, 2304, 768), dict(bias=True, split_tail=True), (256, 232, 0, 0)),
	("ViT-B/32 fc1: three rounds of 200", (12800, 3072, 768), dict(bias=True, act=ops.ACT_QUICKGELU, split_tail=True), (256, 200, 0, 0)),
	("ViT-B/32 proj: 150 tiles", (12800, 768, 768), dict(kind=R, bias=True, split_tail=True), (256, 152, 0, 0)),
	("ViT-B/32 fc2", (12800, 768, 3072), dict(kind=R, bias=True, split_tail=True), (256, 152, 0, 0)),
	("ViT-L/14 qkv: 12 tail tiles x 4 parts", (65792, 3072, 1024), dict(bias=True, split_tail=True), (256, 256, 4, 12)),
	("ViT-L/14 proj", (65792, 1024, 1024), dict(kind=R, bias=True, split_tail=True), (256, 256, 4, 4)),
	("ViT-L/14 fc1", (65792, 4096, 1024), dict(bias=True, act=ops.ACT_GELU, split_tail=True), (256, 256, 4, 16)),
	("ViT-L/14 fc2: 4 tail tiles x 16 parts", (65792, 1024, 4096), dict(kind=R, bias=True, split_tail=True), (256, 256, 16, 4)),
	("SigLIP B/16 proj: fp32 residual on the 256-wide tile", (50176, 768, 768), dict(kind=R, bias=True, split_tail=True), (256, 200, 0, 0)),
	("SigLIP B/16 fc2", (50176, 768, 3072), dict(kind=R, bias=True, split_tail=True), (256, 200, 0, 0)),
	("text tower fc2 at batch 256: 154 tiles", (19712, 512, 2048), dict(kind=R, bias=True, split_tail=True), (256, 160, 0, 0)),
	("below 144 tiles: the 128 x 128 kernel", (8192, 768, 3072), dict(kind=R, bias=True), (0, 0, 0, 0)),
	("small problem", (700, 580, 128), {}, (0, 0, 0, 0)),
	("decode qkv at beam-10 x 256 rows", (2560, 1536, 512), {}, (0, 0, 0, 0)),
	("K not a multiple of 64", (65536, 1024, 4304), dict(kind=R, bias=True), (0, 0, 0, 0)),
]


@pytest.mark.parametrize("name,shape,kw,want", TABLE, ids=[t[0] for t in TABLE])
def test_tile_policy(name, shape, kw, want):
	got = ops.gemm256_plan(*shape, **kw)
	assert (got["tile"], got["workgroups"], got["tail_parts"], got["tail_tiles"]) == want, got


def test_smaller_cu_budget_changes_the_grid_and_the_tail_plan():
	"""novic_persistent_cus: rounds of that many tiles -- the grid shrinks, the tail is whatever is left behind whole rounds of the budget."""
	prev = ops.persistent_cus(208)
	try:
		assert ops.gemm256_plan(12800, 2304, 768, bias=True)["workgroups"] == 152  # 450 tiles: three rounds of 150
		got = ops.gemm256_plan(65792, 1024, 4096, kind=R, bias=True, split_tail=True)  # 1028 tiles = 4 rounds of 208 + 196: no tail of <= 64 tiles
		assert got["tile"] == 256 and got["tail_parts"] == 0 and got["workgroups"] <= 208
	finally:
		ops.persistent_cus(prev)
	assert ops.persistent_cus() == prev
	# the same budget as a per-call argument (novic_epilogue_t.max_workgroups, ABI 8): the process-wide default stays where it is
	with ops.cu_budget(208):
		assert ops.current_cu_budget() == 208 and ops.persistent_cus() == prev
		assert ops.gemm256_plan(12800, 2304, 768, bias=True)["workgroups"] == 152
		with ops.cu_budget(None):
			assert ops.gemm256_plan(12800, 2304, 768, bias=True)["workgroups"] == 232
	assert ops.current_cu_budget() == prev and ops.gemm256_plan(12800, 2304, 768, bias=True)["workgroups"] == 232


def test_coalescing_is_only_offered_where_it_is_exact():
	"""`NativeViT.ksplit_tail_planned` (round 6, advisor): the coalesced tower launches of `Embedder.inference_image_batches` must not run a K-split tail -- nor the single-batch
	launch they replace -- or an image's embedding depends on the launch it shares.  The measured case (ViT-B/32, four caller batches of 256 on the budget the pipeline
	gives 51 200 rows) is exact; ViT-L/14's 257-token shapes always plan a tail, so they are never coalesced; the half stream plans like the fp32 one."""
	from novic_amd import clip_vit, embedders
	b32 = clip_vit.NativeViT(clip_vit.VIT_B_32, seed=1)
	for half in (False, True):
		b32.half_stream = half
		assert not b32.ksplit_tail_planned(256, embedders.pipeline_budget(256 * 50))
		assert not b32.ksplit_tail_planned(1024, embedders.pipeline_budget(1024 * 50))
		assert b32.ksplit_tail_planned(1024, 184)  # (on another budget the same rows do: the guard asks with the budget of the launch)
	l14 = clip_vit.NativeViT(clip_vit.ViTConfig(224, 14, 1024, 1, 16, 4.0, 768))
	assert all(l14.ksplit_tail_planned(n, 256) for n in (64, 128, 256))
	# the half-stream epilogue plans on the 256-wide tile only
	assert ops.gemm256_plan(12800, 768, 768, kind=ops.EPI_RESID_F16, bias=True, split_tail=True)["tile"] == 256
	assert ops.gemm256_plan(12800, 768, 3072, kind=ops.EPI_RESID_F16, bias=True, split_tail=True)["tile"] == 256
	prev = ops.gemm256_pipeline(0)
	try:
		assert ops.gemm256_plan(12800, 768, 768, kind=ops.EPI_RESID_F16, bias=True)["tile"] == 256 and ops.gemm256_plan(12800, 768, 768, kind=R, bias=True)["tile"] == 192
	finally:
		ops.gemm256_pipeline(prev)
