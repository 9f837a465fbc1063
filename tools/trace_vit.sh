#!/bin/bash
# Kernel trace of a ViT tower at batch 256 (run on the GPU box): bash tools/trace_vit.sh [VIT_B_32|VIT_L_14|VIT_H_14] [batch] [half_stream 0|1]
R=${GRAFT_REPO_ROOT:-$(pwd)}
CFG=${1:-VIT_B_32}
BATCH=${2:-256}
HALF=${3:-0}
OUT=$R/gpurun_out/trace_vit
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/vit_run.py <<PY
import sys, time, torch
sys.path.insert(0, "$R")
from novic_amd import clip_vit
vit = clip_vit.NativeViT(clip_vit.$CFG, seed=3).cuda()
vit.half_stream = bool($HALF)
x = torch.randn($BATCH, 3, 224, 224).cuda()
with torch.no_grad():
    for _ in range(3): vit(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): vit(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"$CFG: {dt*1e3:.2f} ms per $BATCH images, {$BATCH/dt:.0f} img/s, {$BATCH/dt*clip_vit.$CFG.flops_per_image()/2.5e15:.3f} of the bf16 MFMA peak")
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/vit_run.py > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60], r["Grid_Size_X"])
    a = agg.setdefault(k, [0, 0]); a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = sum(t for t, c in agg.values())
for (n, gx), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"{n:60s} {gx:>8s} calls {c:5d}  {100 * t / tot:5.1f}%  avg {t / c / 1e3:7.2f} us")
print("total kernel time per forward %.2f ms" % (tot / 8e6))  # 3 warm-up + 5 timed forwards
PY
grep "img/s" $OUT/log.txt
