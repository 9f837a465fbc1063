#!/bin/bash
# Run ON THE GPU BOX: one PMC pass of the training step with the SQ wave-state counters (what bounds each kernel: issue, waits, VALU share).
#   /usr/local/graft/bin/gpurun --timeout 600 -- 'bash tools/collect_sq.sh r02'
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG/sq
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dense --no-decode > $OUT/log.txt 2>&1
echo "sq rc=$?"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
f = sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv"))[-1]
acc = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    k = (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:52], r["Grid_Size"])
    a = acc.setdefault(k, collections.defaultdict(float)); a[r["Counter_Name"]] += float(r["Counter_Value"]); a["n"] += 1
print(f"{'kernel':52s} {'grid':>8s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'valu/active':>11s} {'valu_insts/launch':>17s}")
for (n, g), a in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:24]:
    wc = a["SQ_WAVE_CYCLES"] or 1
    launches = a["n"] / 7
    print(f"{n:52s} {g:>8s} {a['SQ_WAIT_ANY']/wc:8.2f} {a['SQ_WAIT_INST_ANY']/wc:9.2f} {a['SQ_ACTIVE_INST_ANY']/wc:7.2f} {a['SQ_ACTIVE_INST_VALU']/max(a['SQ_ACTIVE_INST_ANY'],1):11.2f} {a['SQ_INSTS_VALU']/max(launches,1):17.0f}")
PY
