"""The C-ABI library loads (no GPU needed) and exports every symbol include/novic_hip.h declares; no compute calls here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
	text = open(os.path.join(ROOT, "include", "novic_hip.h")).read()
	text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
	return sorted(set(re.findall(r"\b(novic_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
	from novic_amd import _lib
	if not os.path.exists(_lib.LIB_PATH):
		_lib.build()
	lib = ctypes.CDLL(_lib.LIB_PATH)
	names = declared_symbols()
	assert len(names) >= 20
	for n in names:
		assert hasattr(lib, n), f"{n} declared in novic_hip.h but not exported"
	assert _lib.lib().novic_abi_version() == _lib.ABI_VERSION == 12
	assert isinstance(_lib.lib().novic_last_error(), bytes)


def test_epilogue_struct_layout_matches_header():
	"""ctypes mirror of novic_epilogue_t (ABI 9): struct_bytes + 2 ints + max_workgroups, 4 pointers, 2 ints, 2 floats, 4 uint32, row_limit, splitk_ws + size = 104 bytes
	on LP64 (ABI 8 carried 40 more: the operands of the LayerNorm fold, a round-4 experiment removed in round 5)."""
	from novic_amd._lib import Epilogue
	assert ctypes.sizeof(Epilogue) == 104 and Epilogue.row_limit.offset == 80 and Epilogue.splitk_ws.offset == 88 and Epilogue.splitk_ws_bytes.offset == 96
	assert Epilogue.struct_bytes.offset == 0 and Epilogue.kind.offset == 4 and Epilogue.c.offset == 16 and Epilogue.ldc.offset == 48 and Epilogue.alpha.offset == 56
	assert Epilogue.seed_lo.offset == 64
	assert _c_struct_size("novic_epilogue_t") == ctypes.sizeof(Epilogue), "ctypes mirror and the header disagree (compiled with gcc)"


def _c_struct_size(name: str, extra: str = "") -> int:
	"""sizeof(name) as gcc sees include/novic_hip.h (the header is plain C by contract)."""
	import subprocess
	import tempfile
	with tempfile.TemporaryDirectory() as d:
		src = os.path.join(d, "s.c")
		open(src, "w").write(f'#include <stdio.h>\n#include "novic_hip.h"\nint main(void) {{ printf("%zu", sizeof({name})); return 0; }}\n')
		exe = os.path.join(d, "s")
		subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), src, "-o", exe], check=True)
		return int(subprocess.run([exe], check=True, capture_output=True, text=True).stdout)


def _doc_struct(name: str):
	"""The ctypes.Structure named `name` as INTEGRATION.md documents it (the class body is executed as written there)."""
	text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
	m = re.search(r"^class " + name + r"\(ctypes\.Structure\):.*?\n(?=\S)", text, flags=re.S | re.M)
	assert m, f"INTEGRATION.md no longer documents {name}"
	ns = {"ctypes": ctypes}
	exec(m.group(0), ns)
	return ns[name]


def test_integration_md_structs_match_the_header():
	"""A binding copied from INTEGRATION.md must have the header's layout (the doc once showed the 80-byte ABI-4 struct)."""
	from novic_amd._lib import AdamWHyper, Epilogue
	doc = _doc_struct("Epilogue")
	assert ctypes.sizeof(doc) == ctypes.sizeof(Epilogue) == _c_struct_size("novic_epilogue_t")
	assert [(n, getattr(doc, n).offset) for n, _ in doc._fields_] == [(n, getattr(Epilogue, n).offset) for n, _ in Epilogue._fields_]
	hyper = _doc_struct("AdamWHyper")
	assert ctypes.sizeof(hyper) == ctypes.sizeof(AdamWHyper) == _c_struct_size("novic_adamw_hyper_t") == 32
	text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
	assert f"`NOVIC_ABI_VERSION` ({_lib_abi()})" in text


def _lib_abi() -> int:
	from novic_amd import _lib
	return _lib.ABI_VERSION


def test_product_path_never_imports_the_oracle():
	pkg = os.path.join(ROOT, "novic_amd")
	for dirpath, _, files in os.walk(pkg):
		for f in files:
			if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
				src = open(os.path.join(dirpath, f)).read()
				assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"
	# tools/ are measurement helpers around the product: they must not reach for the oracle (or the tests' builders) either
	for f in os.listdir(os.path.join(ROOT, "tools")):
		if f.endswith((".py", ".sh")):
			src = open(os.path.join(ROOT, "tools", f)).read()
			assert not re.search(r"^\s*(from|import)\s+(oracle|helpers)\b", src, flags=re.M), f"tools/{f} imports the oracle / test helpers"
	# bench.py: only inside the cpu_baseline leg
	src = open(os.path.join(ROOT, "bench.py")).read()
	for m in re.finditer(r"^\s*(from|import)\s+oracle\b.*$", src, flags=re.M):
		head = src[:m.start()]
		assert head.rfind("def cpu_baseline") > max(head.rfind("\ndef main"), head.rfind("\ndef measure_")), "bench.py imports the oracle outside cpu_baseline()"


def test_cpu_tensors_are_rejected_not_emulated():
	import pytest
	import torch
	from novic_amd import _lib, ops
	a = torch.zeros(8, 8, dtype=torch.bfloat16)
	with pytest.raises(_lib.NovicHipError):
		ops.gemm(a, a, 8, 8, 8, out=torch.zeros(8, 8, dtype=torch.bfloat16))


def test_library_keeps_no_unsynchronised_mutable_state():
	"""include/novic_hip.h, "Process-wide settings": everything the library keeps outside the caller's buffers is thread-local or ONE std::atomic read once per call.
	Greps csrc/ for file-scope / function-local mutable statics (and `g_` globals) and checks each against that rule and against the list the header prints."""
	csrc = os.path.join(ROOT, "novic_amd", "csrc")
	allowed = {"g_err", "g_last_tile",                                                                     # thread_local
	           "g_tile_policy", "g_pipelined", "g_wgrad_pipelined", "g_skinny_wide", "g_attn_policy", "g_beam_step_generic",  # the SIX kernel-selection switches tests pin kernels with (each side is a shipped path)
	           "g_ncu",                                                                                    # default of novic_epilogue_t.max_workgroups
	           "g_tile_counts", "g_trace", "g_trace128",                                                   # diagnostics
	           "attr", "attr_done", "attr_p", "resident"}                                                            # one-time hipFuncSetAttribute flags
	found = set()
	for f in sorted(os.listdir(csrc)):
		if not f.endswith((".hip", ".hpp", ".cpp")):
			continue
		text = re.sub(r"//.*", "", open(os.path.join(csrc, f)).read())
		for ln in text.splitlines():
			s = ln.strip().rstrip("\\").strip()
			m = re.match(r"(?:static\s+)?(?:thread_local\s+)?([\w:<>\*\s]+?)\s*\b(g_\w+)\s*(?:\[\d*\])?\s*(?:[{=;])", s) if re.match(r"(static\s|thread_local\s|std::atomic|int\s|unsigned\s|bool\s|float\s)", s) else None
			if m and "(" not in s.split(m.group(2))[0]:
				found.add(m.group(2))
				assert "std::atomic" in s or "thread_local" in s, f"{f}: `{s}` is process-wide mutable state without synchronisation"
			if re.match(r"static\s", s) and not re.match(r"static\s+(constexpr|const|inline|__device__|__global__|__host__)\b", s) and not re.match(r"static\s+[\w:<>\*\s&]+\(", s):
				name = re.match(r"static\s+(?:thread_local\s+)?(?:std::atomic<[^>]+>|[\w:]+(?:\s+[\w:]+)*?\s*\**)\s+(\w+)\s*[\[{=;]", s)
				assert name, f"{f}: cannot parse `{s}`"
				found.add(name.group(1))
				assert "std::atomic" in s or "thread_local" in s, f"{f}: `{s}` is a mutable static without synchronisation"
	assert found <= allowed, f"mutable state not listed in include/novic_hip.h / this test: {sorted(found - allowed)}"
	assert {"g_ncu", "g_tile_policy", "g_err"} <= found, "the scan no longer sees the known settings: fix its patterns"
	header = open(os.path.join(ROOT, "include", "novic_hip.h")).read()
	for api in ("novic_gemm_tile_policy", "novic_gemm256_pipeline", "novic_wgrad_policy", "novic_skinny_wide_policy", "novic_vit_attn_policy", "novic_beam_step_policy",
	            "novic_persistent_cus", "novic_gemm_tile_counts", "novic_gemm256_trace", "novic_gemm128_trace", "novic_gemm_last_tile", "novic_last_error"):
		assert api in header.split("/* Process-wide settings", 1)[1].split("*/", 1)[0], f"{api} missing from the header's list of process-wide settings"
	assert "no global state except" not in header
