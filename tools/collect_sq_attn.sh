#!/bin/bash
# Run ON THE GPU BOX: SQ wave-state counters of the tower attention kernels at the released shapes (tools/attn_bench.py):  bash tools/collect_sq_attn.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_attn
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $OUT/a -- python3 $R/tools/attn_bench.py > $OUT/log_a.txt 2>&1
echo "pass a rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $OUT/b -- python3 $R/tools/attn_bench.py > $OUT/log_b.txt 2>&1
echo "pass b rc=$?"
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
for sub in ("a", "b"):
    fs = sorted(glob.glob(sys.argv[1] + f"/{sub}/*/*_counter_collection.csv"))
    if not fs:
        print("no counters in pass", sub); continue
    acc = collections.OrderedDict()
    for r in csv.DictReader(open(fs[-1])):
        if "attn" not in r["Kernel_Name"]:
            continue
        k = (re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:44], r["Grid_Size"])
        a = acc.setdefault(k, collections.defaultdict(float)); a[r["Counter_Name"]] += float(r["Counter_Value"])
    names = sorted({c for a in acc.values() for c in a})
    print("pass", sub, names)
    for (n, g), a in acc.items():
        print(f"  {n:44s} {g:>9s} " + " ".join(f"{a[c]:.3e}" for c in names))
PY
