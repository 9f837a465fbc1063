#!/bin/bash
# The decoder attention kernels at REDUCED residency (round 6): what would a fusion cost that parks a weight slice or a row tile in the workgroup's LDS?
#   build (here, no GPU):  bash tools/attn_occupancy_probe.sh build   -> novic_amd/lib/diag/libnovic_hip_attnlds{16,48,112}.so  (attention.hip with that many KiB of unused dynamic LDS
#                                                                        per workgroup: 3 / 2 / 1 workgroups = 12 / 8 / 4 waves per CU instead of 5 / 20)
#   run (GPU):             bash tools/attn_occupancy_probe.sh run     -> gpurun_out/r6_attn_occupancy.txt: dec_attn_fwd / _bwd at the training step's shape under each build
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/novic_amd/csrc; OUT=$ROOT/novic_amd/lib/diag
case "$1" in
build)
	make -C "$CSRC" -j4 >/dev/null
	mkdir -p "$OUT"
	for kb in 16 48 112; do
		/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I"$CSRC" -Wall -Wno-unused-function -ffp-contract=fast -DDEC_ATTN_DIAG_EXTRA_LDS=$((kb * 1024)) \
			-c "$CSRC/attention.hip" -o "$OUT/attention_lds$kb.o"
		objs=$(ls "$CSRC"/build/*.o | grep -v attention.hip.o)
		/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libnovic_hip_attnlds$kb.so" $objs "$OUT/attention_lds$kb.o"
		echo "built $OUT/libnovic_hip_attnlds$kb.so"
	done
	;;
run)
	mkdir -p "$ROOT/gpurun_out"
	cd "$ROOT"
	: > gpurun_out/r6_attn_occupancy.txt
	python tools/attn_occupancy_probe.py 0 >> gpurun_out/r6_attn_occupancy.txt 2>&1
	for kb in 16 48 112; do
		NOVIC_HIP_LIB=$OUT/libnovic_hip_attnlds$kb.so python tools/attn_occupancy_probe.py $kb >> gpurun_out/r6_attn_occupancy.txt 2>&1
	done
	cat gpurun_out/r6_attn_occupancy.txt
	;;
*) echo "usage: $0 build|run"; exit 2;;
esac
