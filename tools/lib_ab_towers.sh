#!/bin/bash
# Two builds of the library against each other on the towers and the training step, alternating, each measurement in its own process: bash tools/lib_ab_towers.sh <other .so> [rounds]
set -e
OTHER=$(realpath "$1"); ROUNDS=${2:-3}
for r in $(seq "$ROUNDS"); do
	for v in default other; do
		if [ $v = other ]; then export NOVIC_HIP_LIB=$OTHER; else unset NOVIC_HIP_LIB; fi
		echo "== round $r: $v"
		python tools/inplace_ab.py VIT_B_32 256 share_buffers | grep "=False"
		python tools/inplace_ab.py VIT_L_14 256 share_buffers | grep "=False"
		python tools/step_ab.py pipeline 1 1 | tail -1
	done
done
