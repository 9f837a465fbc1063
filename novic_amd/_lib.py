"""ctypes loader for libnovic_hip.so (the C ABI of include/novic_hip.h) and the in-tree build driver."""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# $NOVIC_HIP_LIB: another build of the same sources (the A/B tools compare two builds of one kernel in separate processes); it must exist and carry this ABI version
LIB_PATH = os.environ.get("NOVIC_HIP_LIB") or os.path.join(_HERE, "lib", "libnovic_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

_lock = threading.Lock()
_lib = None


class NovicHipError(RuntimeError):
	pass


def build(verbose: bool = False, jobs: int = 4) -> str:
	"""Compile every HIP source for gfx950 into novic_amd/lib/libnovic_hip.so (hipcc cross-compiles without a GPU)."""
	cmd = ["make", "-C", CSRC_DIR, f"-j{jobs}"]
	res = subprocess.run(cmd, capture_output=True, text=True)
	if verbose or res.returncode != 0:
		print(res.stdout[-4000:])
		print(res.stderr[-8000:])
	if res.returncode != 0:
		raise NovicHipError(f"building libnovic_hip.so failed (exit {res.returncode})")
	return LIB_PATH


def build_diag(verbose: bool = False) -> list:
	"""The timing-perturbation builds of the two staggered-wave-group K loops (tools/wgrad_diag.sh build-test: wgrad.hip -DWGRAD_DIAG=1, gemm256.hip -DGEMM256_DIAG_JITTER=1) into
	novic_amd/lib/diag/ -- test instruments, loaded through $NOVIC_HIP_LIB by tests/test_gpu_schedule_diag.py in a process of their own; never the library the package loads."""
	script = os.path.join(os.path.dirname(_HERE), "tools", "wgrad_diag.sh")
	res = subprocess.run(["bash", script, "build-test"], capture_output=True, text=True)
	if verbose or res.returncode != 0:
		print(res.stdout[-2000:])
		print(res.stderr[-4000:])
	if res.returncode != 0:
		raise NovicHipError(f"building the diagnostic libraries failed (exit {res.returncode})")
	d = os.path.join(_HERE, "lib", "diag")
	return sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith(".so"))


class Epilogue(ctypes.Structure):
	"""novic_epilogue_t"""
	_fields_ = [
		("struct_bytes", ctypes.c_uint32), ("kind", ctypes.c_int32), ("act", ctypes.c_int32), ("max_workgroups", ctypes.c_uint32),
		("c", ctypes.c_void_p), ("c2", ctypes.c_void_p), ("resid", ctypes.c_void_p), ("bias", ctypes.c_void_p),
		("ldc", ctypes.c_int32), ("ldr", ctypes.c_int32),
		("alpha", ctypes.c_float), ("drop_p", ctypes.c_float),
		("seed_lo", ctypes.c_uint32), ("seed_hi", ctypes.c_uint32), ("drop_site", ctypes.c_uint32), ("reserved0", ctypes.c_uint32),
		("row_limit", ctypes.c_void_p),
		("splitk_ws", ctypes.c_void_p), ("splitk_ws_bytes", ctypes.c_uint64),
	]


class AdamWHyper(ctypes.Structure):
	"""novic_adamw_hyper_t (host struct, passed to the kernel by value)"""
	_fields_ = [(n, ctypes.c_float) for n in ("lr", "beta1", "beta2", "eps", "weight_decay", "bias_corr1", "bias_corr2", "max_norm")]


class PixelNorm(ctypes.Structure):
	"""novic_pixel_norm_t (by value): the mean / std of the image transform's Normalize step, applied by novic_vit_im2col_u8"""
	_fields_ = [("mean", ctypes.c_float * 3), ("std", ctypes.c_float * 3)]


class WgradProblem(ctypes.Structure):
	"""novic_wgrad_problem_t: one weight gradient dW[M][ldw] (fp32) += dY^T X of a novic_wgradn_bf16 launch"""
	_fields_ = [("dY", ctypes.c_void_p), ("X", ctypes.c_void_p), ("dW", ctypes.c_void_p), ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("ldy", ctypes.c_int32), ("ldx", ctypes.c_int32),
	            ("ldw", ctypes.c_int32), ("reserved0", ctypes.c_int32)]


class NextEmbed(ctypes.Structure):
	"""novic_next_embed_t: the next decode step's inputs as an extra output of a greedy / beam step"""
	_fields_ = [("struct_bytes", ctypes.c_uint32), ("E", ctypes.c_int32), ("wtok", ctypes.c_void_p), ("pos_row", ctypes.c_void_p), ("x_next", ctypes.c_void_p),
	            ("origin_in", ctypes.c_void_p), ("origin_out", ctypes.c_void_p), ("npos", ctypes.c_int32), ("_pad0", ctypes.c_int32)]


ABI_VERSION = 12  # include/novic_hip.h NOVIC_ABI_VERSION


def lib() -> ctypes.CDLL:
	"""The loaded library.  Raises (never falls back) when it has not been built."""
	global _lib
	if _lib is None:
		with _lock:
			if _lib is None:
				if not os.path.exists(LIB_PATH):
					raise NovicHipError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C novic_amd/csrc)")
				handle = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
				handle.novic_last_error.restype = ctypes.c_char_p
				handle.novic_abi_version.restype = ctypes.c_int
				if handle.novic_abi_version() != ABI_VERSION:
					raise NovicHipError(f"{LIB_PATH} has ABI version {handle.novic_abi_version()}, this package needs {ABI_VERSION}: rebuild it (make -C novic_amd/csrc)")
				_lib = handle
	return _lib


def check(rc: int, what: str):
	if rc != 0:
		raise NovicHipError(f"{what} failed with code {rc}: {lib().novic_last_error().decode(errors='replace')}")
