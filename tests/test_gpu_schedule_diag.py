"""The staggered-wave-group K loops (wgrad256p_kernel, gemm256p_kernel) claim that their results depend on barrier and wait COUNTS only, never on when a wave reaches a
segment.  Builds of the two kernels in which every wave draws pseudo-random s_sleep delays (0 / 64 / 256 / ~1000 cycles; a phase is ~310) at every segment boundary
(novic_amd/lib/diag/, built by __graft_entry__.build() through tools/wgrad_diag.sh; csrc/wgrad.hip WGRAD_DIAG, csrc/gemm256.hip GEMM256_DIAG_JITTER) must therefore pass the
bit-identity tests against the one-barrier kernels unchanged.  Each build runs in a process of its own ($NOVIC_HIP_LIB selects the library when novic_amd._lib is imported).
Round 5: written after the one unexplained mismatch of round 4 ("Round 5")."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIAG = os.path.join(ROOT, "novic_amd", "lib", "diag")


@pytest.mark.parametrize("lib,select", [
	("libnovic_hip_wgdiag1.so", "wgrad_8phase or wgrad_pair"),
	("libnovic_hip_g256jit.so", "8phase_gemm or large_tile or k_split_tail or device_row_count or residual_epilogue"),
])
def test_k_loops_are_bit_identical_under_per_wave_timing_perturbation(lib, select):
	path = os.path.join(DIAG, lib)
	if not os.path.exists(path):
		pytest.skip(f"{path} is not built (python -c 'import __graft_entry__ as g; g.build()' builds it)")
	env = dict(os.environ, NOVIC_HIP_LIB=path)
	r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_gemm.py"), "-q", "-m", "gpu", "-x", "-k", select, "-p", "no:cacheprovider"],
	                   env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
	assert " passed" in r.stdout and "failed" not in r.stdout
