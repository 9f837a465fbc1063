"""Per-GEMM / per-kernel times of the last forward in the newest tools/trace_vit.sh trace: python tools/vit_shapes.py (reads gpurun_out/trace_vit)."""
import csv, glob, os, re
f = max(glob.glob('gpurun_out/trace_vit/*/*_kernel_trace.csv'), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = [(re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name']), (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in rows]
L = sum(1 for n, _ in seq if 'vit_attn' in n) // sum(1 for n, _ in seq if 'im2col' in n)
start = [i for i, (n, _) in enumerate(seq) if 'vit_embed' in n][-1] + 1
per, gemm_no = {}, 0
names = ['qkv', 'proj', 'fc1', 'fc2']
for n, d in seq[start:]:
	if 'layernorm' in n:
		key = 'ln'
	elif 'attn' in n:
		key = 'attn'
	elif 'tail_kernel' in n:
		key = names[(gemm_no - 1) % 4] + ' tail'
	elif 'gemm' in n:
		if gemm_no >= 4 * L:
			break
		key = names[gemm_no % 4]
		gemm_no += 1
	else:
		continue
	per.setdefault(key, []).append(d)
for k, v in per.items():
	print(f"{k:10s} {sum(v) / L:8.1f} us per layer  ({len(v) // L} launch(es) of {sum(v) / len(v):7.1f} us)")
print("layers total %.2f ms" % (sum(sum(v) for v in per.values()) / 1e3))
