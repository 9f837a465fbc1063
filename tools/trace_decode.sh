#!/bin/bash
# Kernel trace of decoding (run on the GPU box): bash tools/trace_decode.sh [B] [greedy,beam4,beam10g]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_decode
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/decode_bench.py 10 ${1:-256} ${2:-greedy,beam4} > $OUT/log.txt 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
agg = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60], r["Grid_Size_X"])
    a = agg.setdefault(k, [0, 0]); a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); a[1] += 1
tot = sum(t for t, c in agg.values())
for (n, gx), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:24]:
    print(f"{n:60s} {gx:>8s} calls {c:5d}  {100 * t / tot:5.1f}%  avg {t / c / 1e3:7.2f} us")
print("total kernel time %.1f ms" % (tot / 1e6))
PY
grep "labels/s" $OUT/log.txt
