#!/bin/bash
# Kernel trace of a released checkpoint's image tower at batch 256 (run on the GPU box): bash tools/trace_siglip.sh [b16|so400m|h14_378] [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
WHICH=${1:-b16}
BATCH=${2:-256}
LANES=${3:-0}   # 1: the tower on one stream (kernel durations are not inflated by the other lane's grids)
OUT=$R/gpurun_out/trace_siglip
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/siglip_run.py <<PY
import sys, time, torch
sys.path.insert(0, "$R")
from novic_amd import siglip, clip_vit
if "$WHICH" == "h14_378":
    cfg = clip_vit.ViTConfig(378, 14, 1280, 32, 16, 4.0, 1024, quick_gelu=True)
    vit = clip_vit.NativeViT(cfg, seed=3).cuda()
else:
    cfg = siglip.SigLIPVisionConfig(224, 16, 768, 12, 12, 3072) if "$WHICH" == "b16" else siglip.SigLIPVisionConfig(224, 14, 1152, 27, 16, 4304)
    vit = siglip.NativeSigLIPViT(cfg, seed=3).cuda()
x = torch.randn($BATCH, 3, cfg.image_size, cfg.image_size).cuda()
if $LANES: vit.lanes = $LANES
with torch.no_grad():
    for _ in range(3): vit(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(3): vit(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"$WHICH: {dt*1e3:.2f} ms per $BATCH images, {$BATCH/dt:.0f} img/s, {$BATCH/dt*cfg.flops_per_image()/2.5e15:.3f} of the bf16 MFMA peak")
PY
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 /tmp/siglip_run.py > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "im2col" in r["Kernel_Name"]]
rows = rows[idx[-2]:idx[-1]] if idx[-1] - idx[-2] > 4 else rows[idx[-4]:idx[-2]]
agg = collections.OrderedDict()
for r in rows:
    k = (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60], r["Grid_Size_X"])
    a = agg.setdefault(k, [0, 0]); a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = sum(t for t, c in agg.values())
for (n, gx), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:16]:
    print(f"{n:60s} {gx:>8s} calls {c:5d}  {100 * t / tot:5.1f}%  avg {t / c / 1e3:7.2f} us")
print("kernel time of the last forward %.2f ms; span %.2f ms" % (tot / 1e6, (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e6))
# the first layer's launches in order
for r in rows[2:12]:
    print("   %-60s %8.1f us" % (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
PY
grep "img/s" $OUT/log.txt
