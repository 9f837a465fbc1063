#!/usr/bin/env python3
"""Which of the first HIP streams a process creates share a hardware queue: a spin kernel on stream i and one on stream j take ~1 x its duration when the streams sit on
different queues and ~2 x when they share one.  Prints the pairs that serialise (and each stream against the default stream).  python tools/stream_queue_map.py [n_streams]"""
import os
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
torch.cuda.init()
torch.zeros(1, device="cuda")
streams = [torch.cuda.Stream() for _ in range(n)]
CYC = 400000  # ~0.2 ms of spinning
for s in streams:
	with torch.cuda.stream(s):
		torch.cuda._sleep(1000)
torch.cuda.synchronize()


def timed(a, b):
	torch.cuda.synchronize()
	t0 = time.perf_counter()
	for s in (a, b):
		if s is None:
			torch.cuda._sleep(CYC)
		else:
			with torch.cuda.stream(s):
				torch.cuda._sleep(CYC)
	torch.cuda.synchronize()
	return time.perf_counter() - t0


base = min(timed(streams[0], streams[0]) for _ in range(3)) / 2
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}; one spin kernel {base * 1e3:.3f} ms")
shared = []
for i in range(n):
	t = min(timed(None, streams[i]) for _ in range(2))
	if t > 1.6 * base:
		shared.append(("default", i))
	for j in range(i + 1, n):
		t = min(timed(streams[i], streams[j]) for _ in range(2))
		if t > 1.6 * base:
			shared.append((i, j))
print("pairs that serialise:", shared)
