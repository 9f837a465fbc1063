#!/bin/bash
# Kernel trace of the multiset step (run on the GPU box): bash tools/trace_multiset.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_multiset
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/multiset_run.py > $OUT/log.txt 2>&1
grep -v amdgpu.ids $OUT/log.txt | tail -2
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r".anonymous namespace.::", "", r["Kernel_Name"])
idx = [i for i, r in enumerate(rows) if name(r).startswith("adamw_kernel")]
a, b = idx[-3], idx[-1]  # the last two steps
agg = collections.OrderedDict()
for r in rows[a + 1:b + 1]:
    k = (name(r)[:64], r["Grid_Size_X"])
    x = agg.setdefault(k, [0, 0]); x[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); x[1] += 1
tot = sum(t for t, c in agg.values())
for (n, gx), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:26]:
    print(f"{n:64s} {gx:>8s} launches/step {c / 2:5.1f}  {100 * t / tot:5.1f}%  avg {t / c / 1e3:8.2f} us  per step {t / 2e3:8.1f} us")
print("kernel time per step %.1f us, span %.1f us" % (tot / 2e3, (int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 2e3))
PY
