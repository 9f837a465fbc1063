#!/usr/bin/env python3
"""bench.py's action_train leg alone, for a kernel trace (bash tools/trace_train_loop.sh): python tools/train_loop_run.py [resident | streaming] [steps per chunk]  (default: resident 8)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else "resident"
assert leg in ("resident", "streaming")
spc = int(sys.argv[2]) if len(sys.argv) > 2 else 8  # optimizer steps per chunk (the chunk's one host synchronisation is inside the logged rate)
res = bench.measure_train_loop(torch.device("cuda"), bench.ACCUM, 1.0, legs=(leg,), steps_per_chunk=spc)
print({k: v for k, v in res.items() if k.endswith("samples_per_s")})
