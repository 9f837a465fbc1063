#!/bin/bash
# Gaps between consecutive kernels of a tower forward (run on the GPU box): bash tools/trace_gaps.sh [CFG] [batch]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/vit_gaps.py ${1:-VIT_B_32} ${2:-256} > $OUT/log.txt 2>&1
cat $OUT/log.txt | grep -v amdgpu.ids
python3 - $OUT <<'PY'
import csv, glob, sys, re
rows = sorted(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-120:]  # the last forward (graph replay)
gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(rows, rows[1:])]
dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
print("last 120 kernels: sum of durations %.1f us, sum of gaps %.1f us, span %.1f us" % (sum(dur), sum(gaps), (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3))
for r, d, g in list(zip(rows, dur, gaps + [0]))[30:52]:
    print(f"{re.sub(r'.anonymous namespace.::', '', r['Kernel_Name'])[:48]:48s} grid {r['Grid_Size_X']:>8s} dur {d:7.1f}  gap after {g:6.1f}")
PY
