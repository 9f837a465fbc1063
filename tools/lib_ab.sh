#!/bin/bash
# Two builds of the library against each other, alternating, each measurement in its own process: bash tools/lib_ab.sh <other .so> [rounds]
# (build the variant with e.g. hipcc ... -DGEMM256P_B1_AT_PHASE3=0 -c gemm256.hip and link it beside the other objects; $NOVIC_HIP_LIB selects it: novic_amd/_lib.py)
set -e
OTHER=$(realpath "$1"); ROUNDS=${2:-3}
mkdir -p gpurun_out
for r in $(seq "$ROUNDS"); do
	for v in default other; do
		if [ $v = other ]; then export NOVIC_HIP_LIB=$OTHER; else unset NOVIC_HIP_LIB; fi
		echo "== round $r: $v"
		python tools/gemm_bench.py | grep -E "fwd|bwd|sum"
		python tools/vit_b32_gemm_ab.py | tail -6
		python tools/step_ab.py pipeline 1 1 | tail -1
	done
done
