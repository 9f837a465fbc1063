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
	assert _lib.lib().novic_abi_version() == _lib.ABI_VERSION == 5
	assert isinstance(_lib.lib().novic_last_error(), bytes)


def test_epilogue_struct_layout_matches_header():
	"""ctypes mirror of novic_epilogue_t: 2 ints, 4 pointers, 2 ints, 2 floats, 4 uint32, row_limit, splitk_ws + size = 96 bytes on LP64."""
	from novic_amd._lib import Epilogue
	assert ctypes.sizeof(Epilogue) == 96 and Epilogue.row_limit.offset == 72 and Epilogue.splitk_ws.offset == 80 and Epilogue.splitk_ws_bytes.offset == 88
	assert Epilogue.c.offset == 8 and Epilogue.ldc.offset == 40 and Epilogue.alpha.offset == 48 and Epilogue.seed_lo.offset == 56


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
