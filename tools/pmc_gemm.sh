#!/bin/bash
# PMC passes over the GEMM sweep tool (run on the GPU box): bash tools/pmc_gemm.sh "<shape>" COUNTER [COUNTER ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
SHAPE=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
	rm -rf /tmp/pmc_$c
	timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/tools/gemm_sweep.py $SHAPE > /tmp/pmc_$c.log 2>&1
	python3 - "$c" <<'PY'
import csv, glob, sys, collections
c = sys.argv[1]
f = glob.glob(f"/tmp/pmc_{c}/*/*_counter_collection.csv")
if not f:
    print(c, "no output"); sys.exit(0)
agg = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f[0])):
    if "gemm" in r["Kernel_Name"]:
        k = ("256" if "gemm256" in r["Kernel_Name"] else "128") + " " + r["Counter_Name"]
        agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
for k, (v, n) in sorted(agg.items()):
    print(f"{k:40s} per launch {v / n:16.1f}  ({n} launches)")
PY
done
