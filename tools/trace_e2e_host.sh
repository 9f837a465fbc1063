#!/bin/bash
# Kernel + memory-copy trace of the coalesced pipeline fed from pinned host uint8 batches with two decode lanes, for GPU_MAX_HW_QUEUES unset and 4 (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp E2E_LONG=6 E2E_QUIET=1
for q in unset 4; do
  OUT=$R/gpurun_out/trace_e2e_host_$q
  rm -rf $OUT; mkdir -p $OUT
  if [ $q = unset ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT -- python3 $R/tools/e2e_timeline.py 4 1024 208 1 small 2 host_u8 > $OUT/log.txt 2>&1
  echo "== GPU_MAX_HW_QUEUES=$q"; grep unrecorded $OUT/log.txt
  python3 - $OUT <<'PY'
import csv, glob, sys, collections, os
d = sys.argv[1]
kt = sorted(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
mc = sorted(glob.glob(d + "/*/*_memory_copy_trace.csv"), key=os.path.getmtime)
rows = list(csv.DictReader(open(kt)))
qs = collections.Counter((r["Queue_Id"], r["Stream_Id"]) for r in rows)
print("kernel launches per (queue, stream):", dict(qs))
if mc:
    cp = list(csv.DictReader(open(mc[-1])))
    print("memory copy columns:", list(cp[0].keys()) if cp else None, "copies:", len(cp))
    big = [c for c in cp if int(c.get("End_Timestamp", 0)) - int(c.get("Start_Timestamp", 0)) > 200_000]
    by = collections.Counter((c.get("Direction"), c.get("Stream_Id")) for c in cp)
    print("copies by (direction, stream):", dict(by))
    ds = sorted(int(c["End_Timestamp"]) - int(c["Start_Timestamp"]) for c in big)
    if ds:
        print(f"copies longer than 0.2 ms: {len(ds)}, median {ds[len(ds)//2]/1e3:.0f} us, max {ds[-1]/1e3:.0f} us")
PY
done
