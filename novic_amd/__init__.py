"""novic_amd -- MI355X-native implementation of NOVIC's CLIP-embedding -> label-decoder hot path.

The arithmetic lives in hand-written HIP kernels for gfx950 (``novic_amd/csrc``) behind the C ABI declared in
``include/novic_hip.h`` (``novic_amd/lib/libnovic_hip.so``).  The Python modules here mirror the reference's
call surface (``embedding_decoder``, ``embedding_noise``, ``embedders``, ``embedding_dataset``, ``infer``, ``train``).
There is no CPU or eager-PyTorch fallback: a missing library or a missing GPU raises.
"""
__version__ = "0.1.0"
