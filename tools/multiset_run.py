#!/usr/bin/env python3
"""bench.py's configs[4] leg alone (multiset fine-tune step: F = 1024, three weighted targets per embedding), for a kernel trace: bash tools/trace_multiset.sh"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

print(bench.measure_multiset(torch.device("cuda"), 0, 1, None, bench.ACCUM, steps=5))
