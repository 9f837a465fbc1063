#!/bin/bash
# Kernel trace of the action_train leg (run on the GPU box): where the loop's optimizer steps differ from the bare step -- bash tools/trace_train_loop.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_train_loop
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $R/tools/train_loop_run.py > $OUT/log.txt 2>&1
grep -v amdgpu.ids $OUT/log.txt | tail -3
python3 - $OUT <<'PY'
import csv, glob, sys, re, collections
rows = sorted(csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r".anonymous namespace.::", "", r["Kernel_Name"])
# optimizer steps = intervals between consecutive adamw launches; take the steps of the resident run's later chunks (the first run in the file)
idx = [i for i, r in enumerate(rows) if name(r).startswith("adamw_kernel")]
idx = idx[:40]  # resident leg: 5 chunks x 8 steps
steps = [(idx[i], idx[i + 1]) for i in range(16, 38)]
busy, span, extra = [], [], collections.Counter()
bigs = []
for a, b in steps:
    seg = rows[a + 1:b + 1]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(seg[-1]["End_Timestamp"])
    # union of kernel intervals (streams overlap)
    iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in seg)
    u, cur_s, cur_e = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s > cur_e:
            u += cur_e - cur_s; cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    u += cur_e - cur_s
    busy.append(u / 1e3); span.append((t1 - t0) / 1e3)
    gaps = sorted(((int(y["Start_Timestamp"]) - int(x["End_Timestamp"])) / 1e3, name(x)[:40], name(y)[:40]) for x, y in zip(seg, seg[1:]))
    bigs.append(gaps[-3:])
    for r in seg:
        extra[name(r)[:70]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
n = len(steps)
print(f"{n} optimizer steps of the loop: span {sum(span) / n:.1f} us per step, GPU busy (union over streams) {sum(busy) / n:.1f} us, idle {sum(span) / n - sum(busy) / n:.1f} us")
print("largest gaps of the last step:", bigs[-1])
print("kernels the loop adds or that run beside the step (us per step):")
for k, v in extra.most_common():
    if any(s in k for s in ("cache_gather", "Cat", "copy", "elementwise", "index", "noise", "fill", "reduce_kernel", "gather")):
        print(f"  {k:70s} {v / n:8.1f}")
PY
