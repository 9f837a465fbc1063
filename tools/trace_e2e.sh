#!/bin/bash
# Kernel trace of the coalesced ViT-B/32 + greedy decode pipeline (run on the GPU box), read PER STREAM: bash tools/trace_e2e.sh [batches per launch:decode rows] [budget]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/trace_e2e
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export E2E_HALF=1 E2E_COALESCE=${1:-4:1024} E2E_SOURCES=resident E2E_GREEDY_ONLY=1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/tools/e2e_coalesce.py ${2:-} > $OUT/log.txt 2>&1
grep "budget" $OUT/log.txt
python3 $R/tools/streams_of_trace.py $OUT
