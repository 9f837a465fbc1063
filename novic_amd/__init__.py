"""novic_amd -- MI355X-native implementation of NOVIC's CLIP-embedding -> label-decoder hot path.

The arithmetic lives in hand-written HIP kernels for gfx950 (``novic_amd/csrc``) behind the C ABI declared in
``include/novic_hip.h`` (``novic_amd/lib/libnovic_hip.so``).  The Python modules here mirror the reference's
call surface (``embedding_decoder``, ``embedding_noise``, ``embedders``, ``embedding_dataset``, ``infer``, ``train``).
There is no CPU or eager-PyTorch fallback: a missing library or a missing GPU raises.
"""
import os as _os

# Concurrent decode lanes / towers run on HIP streams of their own; the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4), and two
# lanes that land on one queue run one after the other: three decode lanes reach 157 k greedy labels/s on 4 queues and 250 k on 8 (tools/decode_bench.py).  The
# variable is read when the HIP runtime starts (the first torch.cuda call), so import this package -- or set it yourself -- before that; a value already set wins.
# NOT in a multi-rank job (WORLD_SIZE > 1): RCCL brings streams and kernels of its own, eight hardware queues beside them have never been run on hardware (this pool
# has no multi-GPU node), and data-parallel training uses no lanes -- a rank that wants decode lanes there sets the variable itself.
if int(_os.environ.get("WORLD_SIZE", "1") or "1") <= 1:
	_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"
