"""Thin torch-tensor -> C-ABI adapters (one function per entry point of include/novic_hip.h).

PyTorch here is plumbing only: it owns device memory and the HIP stream; every FLOP below runs in libnovic_hip.so.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _lib
from ._lib import Epilogue, check

EPI_STORE_BF16, EPI_STORE_F32, EPI_ATOMIC_F32, EPI_RESID_F32, EPI_GELU_BF16, EPI_GELU_BWD_BF16 = range(6)
ACT_NONE, ACT_GELU, ACT_QUICKGELU = range(3)

_vp = ctypes.c_void_p


def _ptr(t: Optional[torch.Tensor]):
	return _vp(0) if t is None else _vp(t.data_ptr())


def _stream():
	return _vp(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
	for t in ts:
		if t is not None and not t.is_cuda:
			raise _lib.NovicHipError("novic_amd kernels need tensors on an MI355X device (no CPU fallback exists)")


class Dropout:
	"""Philox dropout descriptor shared by forward and backward of one site."""
	__slots__ = ("p", "seed", "site")

	def __init__(self, p: float = 0.0, seed: int = 0, site: int = 0):
		self.p, self.seed, self.site = float(p), int(seed), int(site)

	def at(self, site: int) -> "Dropout":
		return Dropout(self.p, self.seed, site)


NO_DROPOUT = Dropout()


def gemm(a: torch.Tensor, b: torch.Tensor, M: int, N: int, K: int, *, a_kstrided=False, b_kstrided=False, kind=EPI_STORE_BF16, out: torch.Tensor,
         out2: Optional[torch.Tensor] = None, resid: Optional[torch.Tensor] = None, bias: Optional[torch.Tensor] = None, act=ACT_NONE,
         alpha: float = 1.0, split_k: int = 1, dropout: Dropout = NO_DROPOUT, lda: Optional[int] = None, ldb: Optional[int] = None,
         ldc: Optional[int] = None, ldr: Optional[int] = None):
	"""C[M,N] = A*B with a fused epilogue (novic_gemm_bf16).  a/b are bf16 2-D tensors in the storage the flags name."""
	_dev(a, b, out)
	assert a.dtype == torch.bfloat16 and b.dtype == torch.bfloat16
	ep = Epilogue()
	ep.kind, ep.act = kind, act
	ep.c, ep.c2, ep.resid, ep.bias = out.data_ptr(), (out2.data_ptr() if out2 is not None else 0), (resid.data_ptr() if resid is not None else 0), (bias.data_ptr() if bias is not None else 0)
	ep.ldc = ldc if ldc is not None else out.stride(-2)
	ep.ldr = ldr if ldr is not None else (resid.stride(-2) if resid is not None else 0)
	ep.alpha, ep.drop_p = alpha, dropout.p
	ep.seed_lo, ep.seed_hi, ep.drop_site = dropout.seed & 0xFFFFFFFF, (dropout.seed >> 32) & 0xFFFFFFFF, dropout.site
	rc = _lib.lib().novic_gemm_bf16(_ptr(a), _ptr(b), M, N, K, lda if lda is not None else a.stride(-2), ldb if ldb is not None else b.stride(-2),
	                                int(a_kstrided), int(b_kstrided), split_k, ctypes.byref(ep), _stream())
	check(rc, "novic_gemm_bf16")
	return out
