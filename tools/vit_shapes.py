"""Per-shape kernel times of the newest tools/trace_vit.sh trace: python tools/vit_shapes.py (reads gpurun_out/trace_vit)."""
import csv, glob, os, re
f = max(glob.glob('gpurun_out/trace_vit/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:44], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
last = len(seq) - 1 - [n for n, _ in seq][::-1].index(next(n for n, _ in seq if 'vit_attn' in n))
L = sum(1 for n, _ in seq if 'vit_attn' in n) // sum(1 for n, _ in seq if 'im2col' in n)
# the last forward: walk back over L layers of (ln, qkv, attn, proj, ln, fc1, fc2)
per = {}
att = [i for i, (n, _) in enumerate(seq) if 'vit_attn' in n][-L:]
for i in att:
	for name, off in (('ln1', -2), ('qkv', -1), ('attn', 0), ('proj', 1), ('ln2', 2), ('fc1', 3), ('fc2', 4)):
		per.setdefault(name, []).append(seq[i + off])
for name, v in per.items():
	print(f"{name:5s} {v[0][0]:46s} avg {sum(d for _, d in v) / len(v):8.1f} us  x {len(v)}")
print("layers total %.2f ms" % (sum(d for v in per.values() for _, d in v) / 1e3))
