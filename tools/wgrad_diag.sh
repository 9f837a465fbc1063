#!/bin/bash
# The deterministic instruments for the 8-phase weight-gradient schedule (novic_amd/csrc/wgrad.hip, WGRAD_DIAG):
#   build:  bash tools/wgrad_diag.sh build      -> novic_amd/lib/diag/libnovic_hip_wgdiag{1,2,3}.so, libnovic_hip_g256jit.so (here, no GPU; the .so travel with the gpurun snapshot)
#   run:    bash tools/wgrad_diag.sh run        -> (GPU) the bit-identity test of tests/test_gpu_gemm.py under both builds, ONCE each:
#             diag1 (pseudo-random per-wave delays at every segment boundary)  must PASS  -- the result depends on barrier / wait counts only, not on timing
#             diag2 (the same + the steady counted waits removed), diag3 (waits removed, no delays): reported -- how much margin the staging distance alone leaves
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CSRC=$ROOT/novic_amd/csrc; OUT=$ROOT/novic_amd/lib/diag
case "$1" in
build|build-test)  # build-test: only the two libraries tests/test_gpu_schedule_diag.py loads (what __graft_entry__.build() asks for)
	make -C "$CSRC" -j4 >/dev/null
	mkdir -p "$OUT"
	if [ "$1" = build-test ]; then  # up to date (newer than every source and than the library they link against): nothing to do
		newest=$(ls -t "$CSRC"/*.hip "$CSRC"/*.hpp "$CSRC"/*.cpp "$ROOT"/include/*.h "$ROOT"/novic_amd/lib/libnovic_hip.so "$0" | head -1)
		if [ -f "$OUT/libnovic_hip_wgdiag1.so" ] && [ -f "$OUT/libnovic_hip_g256jit.so" ] && [ "$OUT/libnovic_hip_wgdiag1.so" -nt "$newest" ] && [ "$OUT/libnovic_hip_g256jit.so" -nt "$newest" ]; then
			echo "diagnostic libraries up to date"; exit 0
		fi
	fi
	for d in $([ "$1" = build-test ] && echo 1 || echo 1 2 3); do
		/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I"$CSRC" -Wall -Wno-unused-function -ffp-contract=fast -DWGRAD_DIAG=$d \
			-c "$CSRC/wgrad.hip" -o "$OUT/wgrad_d$d.o"
		objs=$(ls "$CSRC"/build/*.o | grep -v wgrad.hip.o)
		/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libnovic_hip_wgdiag$d.so" $objs "$OUT/wgrad_d$d.o"
		echo "built $OUT/libnovic_hip_wgdiag$d.so"
	done
	# the forward / input-gradient GEMM's four-phase K loop under the same per-wave delays (gemm256.hip, GEMM256_DIAG_JITTER)
	/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I"$ROOT/include" -I"$CSRC" -Wall -Wno-unused-function -ffp-contract=fast -DGEMM256_DIAG_JITTER=1 \
		-c "$CSRC/gemm256.hip" -o "$OUT/gemm256_jit.o"
	objs=$(ls "$CSRC"/build/*.o | grep -v gemm256.hip.o)
	/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$OUT/libnovic_hip_g256jit.so" $objs "$OUT/gemm256_jit.o"
	echo "built $OUT/libnovic_hip_g256jit.so"
	;;
run)
	mkdir -p "$ROOT/gpurun_out"
	cd "$ROOT"
	rc1=0; rc2=0
	NOVIC_HIP_LIB=$OUT/libnovic_hip_wgdiag1.so timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "wgrad_8phase or wgrad_pair" -x > gpurun_out/r5_wgdiag1.txt 2>&1 || rc1=$?
	tail -3 gpurun_out/r5_wgdiag1.txt
	NOVIC_HIP_LIB=$OUT/libnovic_hip_wgdiag2.so timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "wgrad_8phase" > gpurun_out/r5_wgdiag2.txt 2>&1 || rc2=$?
	tail -3 gpurun_out/r5_wgdiag2.txt
	rc3=0
	NOVIC_HIP_LIB=$OUT/libnovic_hip_wgdiag3.so timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "wgrad_8phase" > gpurun_out/r5_wgdiag3.txt 2>&1 || rc3=$?
	tail -3 gpurun_out/r5_wgdiag3.txt
	rc4=0
	NOVIC_HIP_LIB=$OUT/libnovic_hip_g256jit.so timeout -k 10 600 python -m pytest tests/test_gpu_gemm.py -q -m gpu -k "8phase_gemm or large_tile or k_split_tail or device_row_count or beyond_2_gib or residual_epilogue or forced_256" > gpurun_out/r5_g256jit.txt 2>&1 || rc4=$?
	tail -3 gpurun_out/r5_g256jit.txt
	echo "gemm256p under per-wave delays (must pass) exit $rc4"
	echo "diag1 (jitter; must pass) exit $rc1; diag2 (jitter + steady waits removed) exit $rc2; diag3 (steady waits removed, full speed) exit $rc3"
	[ $rc1 -eq 0 ] && [ $rc4 -eq 0 ]
	;;
*) echo "usage: $0 build|build-test|run"; exit 2;;
esac
