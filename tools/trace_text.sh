#!/bin/bash
# Kernel trace of the CLIP text tower (B/32 width 512, 77 tokens) at batch 256: bash tools/trace_text.sh [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
BATCH=${1:-256}
OUT=$R/gpurun_out/trace_text
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/text_run.py <<PY
import sys, time, torch
sys.path.insert(0, "$R")
from novic_amd import clip_text
cfg = clip_text.TEXT_B_32
tw = clip_text.NativeTextTower(cfg, seed=3).cuda()
g = torch.Generator().manual_seed(1)
ids = torch.randint(1, cfg.vocab_size - 2, ($BATCH, cfg.context_length), generator=g)
ids[:, 0] = cfg.vocab_size - 2
lens = torch.randint(4, cfg.context_length, ($BATCH,), generator=g)
for i in range($BATCH):
    ids[i, lens[i]] = cfg.vocab_size - 1; ids[i, lens[i] + 1:] = 0
ids = ids.cuda()
with torch.no_grad():
    for _ in range(3): tw(ids)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): tw(ids)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"text tower: {dt*1e3:.2f} ms per $BATCH texts, {$BATCH/dt:.0f} texts/s")
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/text_run.py > $OUT/log.txt 2>&1
grep "texts/s" $OUT/log.txt
