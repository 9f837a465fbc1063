#!/usr/bin/env python3
"""How three decode lanes' throughput depends on HOW MANY streams the process created (and used) before the lanes' own: the ROCm runtime maps HIP streams onto
GPU_MAX_HW_QUEUES hardware queues, and lanes that share a queue run one after the other.  python tools/lane_queue_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time, torch
sys.path.insert(0, %r)
import bench
n_before = int(sys.argv[1])
spec = bench.WorkloadSpec(embed_dim=512, vocab_size=6912, token_length=12)
torch.manual_seed(1)
model = bench.build_decoder(spec, dropout=0.0, device=torch.device("cuda"))
with torch.no_grad():
    model.logits_linear.weight[0].zero_()
model.eval()
dummies = [torch.cuda.Stream() for _ in range(n_before)]
for s in dummies:
    with torch.cuda.stream(s):
        torch.zeros(16, device="cuda").add_(1)
torch.cuda.synchronize()
es = [torch.nn.functional.normalize(torch.randn(256, 512), dim=-1).cuda() for _ in range(3)]
with torch.no_grad():
    for _ in range(3):
        model.generate_many(es, False, True, 1.0, 0.0, None, None, False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(8):
        model.generate_many(es, False, True, 1.0, 0.0, None, None, False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 8
print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} streams used before the lanes: {n_before}: 3 x 256 greedy {3 * 256 / dt:.0f} labels/s", flush=True)
''' % ROOT
for n in (0, 1, 2, 3, 4, 5, 6, 7, 8, 12):
	subprocess.run([sys.executable, "-c", CHILD, str(n)], check=False)
