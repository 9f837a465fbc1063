#!/bin/bash
# Run ON THE GPU BOX (via gpurun): kernel trace + the two HBM PMC passes + one MFMA-utilisation PMC pass of the same bench command, each in its own rocprofv3 run.
#   /usr/local/graft/bin/gpurun --timeout 900 -- 'bash tools/collect_profile.sh r01'
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
python3 $R/bench.py --fingerprint > $OUT/source_fingerprint.txt
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 5 --warmup 2 --repeats 1 --no-cpu-baseline --no-dense"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1
echo "trace rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- $CMD --no-decode > $OUT/train.log 2>&1
echo "train-only trace rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- $CMD --no-decode > $OUT/fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- $CMD --no-decode > $OUT/write.log 2>&1
echo "write rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/mfma -- $CMD --no-decode > $OUT/mfma.log 2>&1
echo "mfma rc=$?"
grep -h '^{"metric"' $OUT/trace.log | tail -1 > $OUT/bench_line_under_profiler.json || true
