"""The launches of ONE decode step in stream order, from a rocprofv3 kernel trace of tools/decode_bench.py (run on the GPU box behind tools/trace_decode.sh):
python3 tools/decode_step_sequence.py gpurun_out/trace_decode  -- prints the kernels between two consecutive logits GEMMs late in the trace (a steady step) with start offsets,
durations and the idle gap in front of each, i.e. where a step's time goes kernel by kernel rather than summed over the run."""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])[:70]
# a step ends with its selection kernel (greedy_step / beam_step*); take the steps of the last quarter of the trace
ends = [i for i, r in enumerate(rows) if re.search(r"greedy_step|beam_step", r["Kernel_Name"])]
want = sys.argv[2] if len(sys.argv) > 2 else "greedy_step"
ends = [i for i in ends if want in rows[i]["Kernel_Name"]]
if len(ends) < 4:
	sys.exit("no steps of %s in the trace" % want)
if len(sys.argv) > 3 and sys.argv[3] == "call":  # from the last step of one call through the second step of the next: what lies BETWEEN calls (finish, begin, reset, prefix pass)
	steps_per_call = int(sys.argv[4]) if len(sys.argv) > 4 else 11
	a, b = ends[-steps_per_call - 2], ends[-steps_per_call + 1]
else:
	a, b = ends[-3], ends[-2]
t0 = int(rows[a]["End_Timestamp"])
prev_end = t0
print(f"{'kernel':70s} {'grid':>8s} {'start':>8s} {'dur':>7s} {'gap':>6s}  (us; step = {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches)")
for r in rows[a + 1:b + 1]:
	s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
	print(f"{name(r):70s} {r['Grid_Size_X']:>8s} {(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.2f} {(s - prev_end) / 1e3:6.2f}")
	prev_end = e
