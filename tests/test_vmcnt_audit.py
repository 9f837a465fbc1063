"""Every hand-written counted `s_waitcnt vmcnt(N)` in the HIP sources, audited in the generated gfx950 ISA (tools/audit_vmcnt.py): no flat_* or scratch_*
instruction in a kernel that holds one, straight-line waits preceded by at least N vector-memory issues in their own basic block, cross-block waits only
where the argument is written down.  Compiles the source files that hold such waits to assembly (about a minute); needs hipcc, not a GPU."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") and shutil.which("hipcc") is None, reason="hipcc not available")
def test_counted_waits_hold_in_the_generated_isa(capsys):
	spec = importlib.util.spec_from_file_location("audit_vmcnt", os.path.join(ROOT, "tools", "audit_vmcnt.py"))
	mod = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(mod)
	rc = mod.main([])
	out = capsys.readouterr().out
	assert rc == 0, out
	assert "skinny_n128_kernel" in out and "gemm256_kernel" in out and "violations: 0" in out  # the audit saw the kernels it is there for
	# the 8-phase kernels: a steady loop was found in each, made of LDS-DMA half-tile groups only, every wait leaving whole groups in flight
	assert out.count("gemm256p_kernel") >= 2 and out.count("wgrad256p_kernel") >= 2 and "steady loop" in out and "VIOLATION" not in out


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-readelf"), reason="llvm-readelf not available")
def test_no_kernel_of_the_built_library_uses_scratch_unannounced(capsys):
	"""tools/audit_scratch.py over the library the tests load: a private segment (spills, or an array hipcc chose to index dynamically -- round 4's dec_attn_bwd_kernel:
	1.5 % of the training step, invisible in source, build log and results) only on the kernels listed there with a reason."""
	from novic_amd import _lib
	if not os.path.exists(_lib.LIB_PATH):
		pytest.skip("library not built")
	spec = importlib.util.spec_from_file_location("audit_scratch", os.path.join(ROOT, "tools", "audit_scratch.py"))
	mod = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(mod)
	rc = mod.main([_lib.LIB_PATH])
	out = capsys.readouterr().out
	assert rc == 0, out
	m = __import__("re").search(r"kernels: (\d+)", out)
	assert m and int(m.group(1)) > 200, out  # (the audit really saw the library's kernels)
	assert "dec_attn_bwd_kernelILi64ELi1E" not in out
