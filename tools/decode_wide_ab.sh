#!/bin/bash
# Two builds of the small-tile decode kernels against each other, alternating processes: eight waves per workgroup (128 output columns) where four would need more than one
# round of the chip (decode_fused.hip, DECODE_LN_WIDE = 1: the shipped form) against four everywhere (-DDECODE_LN_WIDE=0).
#   build (here): bash tools/decode_wide_ab.sh build     run (GPU): bash tools/decode_wide_ab.sh run
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); CSRC=$ROOT/novic_amd/csrc; OUT=$ROOT/novic_amd/lib/diag
case "$1" in
build)
	make -C "$CSRC" -j4 >/dev/null; mkdir -p "$OUT"
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I"$CSRC" -Wall -Wno-unused-function -ffp-contract=fast -DDECODE_LN_WIDE=0 -c "$CSRC/decode_fused.hip" -o "$OUT/decode_fused_narrow.o"
	/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libnovic_hip_lnnarrow.so" $(ls "$CSRC"/build/*.o | grep -v decode_fused.hip.o) "$OUT/decode_fused_narrow.o"
	echo "built $OUT/libnovic_hip_lnnarrow.so";;
run)
	cd "$ROOT"
	for r in 1 2 3; do
		for v in wide narrow; do
			if [ $v = narrow ]; then export NOVIC_HIP_LIB=$OUT/libnovic_hip_lnnarrow.so; else unset NOVIC_HIP_LIB; fi
			echo "== round $r: $v"
			python tools/decode_bench.py 12 256 greedy,beam4,beam10g 2>&1 | grep labels
			python tools/decode_bench.py 6 1024 greedy,beam4 2>&1 | grep labels
		done
	done;;
*) echo "usage: $0 build|run"; exit 2;;
esac
