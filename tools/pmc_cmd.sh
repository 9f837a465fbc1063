#!/bin/bash
# One PMC pass per counter over an arbitrary python tool (run on the GPU box): bash tools/pmc_cmd.sh "tools/gemm_bench.py" COUNTER [COUNTER ...]
# Prints the per-launch average of each counter for every (kernel, grid).
R=${GRAFT_REPO_ROOT:-$(pwd)}
CMD=$1; shift
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
	rm -rf /tmp/pmc_$c
	timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/$CMD > /tmp/pmc_$c.log 2>&1
	python3 - "$c" <<'PY'
import csv, glob, sys, collections, re
c = sys.argv[1]
f = glob.glob(f"/tmp/pmc_{c}/*/*_counter_collection.csv")
if not f:
    print(c, "no output"); sys.exit(0)
agg = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    k = (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:60], r["Grid_Size"])
    a = agg.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
print("==", c)
for (k, g), (v, n) in agg.items():
    if v / n > 1000: print(f"{k:60s} grid {g:>8s}  per launch {v / n:14.1f}  ({n})")
PY
done
