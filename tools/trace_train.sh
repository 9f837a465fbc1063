#!/bin/bash
# Kernel trace of the training-only bench (run on the GPU box): bash tools/trace_train.sh  -> prints the per-step breakdown of the top kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_train
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode --no-dense > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70], r["Grid_Size_X"], r["Grid_Size_Z"])
    a = agg.setdefault(k, [0, 0]); a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = 0
for (n, gx, gz), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:26]:
    print(f"{n:70s} {gx:>8s} {gz:>3s} calls {c:4d}  {t / 7e3:8.1f} us/step  avg {t / c / 1e3:7.1f}")
print("sum all kernels / 7 steps: %.1f us" % (sum(t for t, c in agg.values()) / 7e3))
PY
grep '"metric"' $OUT/log.txt | cut -c1-220
