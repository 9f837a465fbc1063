#!/usr/bin/env python3
"""bench.py's action_train leg alone (cache resident in HBM unless argv[1] == 'streaming'), for a kernel trace: bash tools/trace_train_loop.sh"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "streaming":
	os.environ["NOVIC_LOADER_HBM_BUDGET"] = "0"
	bench_orig = os.environ.pop
res = bench.measure_train_loop(torch.device("cuda"), bench.ACCUM, 1.0)
print({k: v for k, v in res.items() if k.endswith("samples_per_s")})
