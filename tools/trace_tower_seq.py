#!/usr/bin/env python3
"""Kernel sequence of the LAST forward in a rocprofv3 kernel trace of a tower run: name, grid, start offset, duration, gap to the previous kernel.
python tools/trace_tower_seq.py TRACE_DIR [first_kernel_substring]"""
import csv, glob, os, re, sys
d = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "im2col"
f = max(glob.glob(d + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if first in r["Kernel_Name"]]
rows = rows[starts[-1]:]
t0, prev_end = int(rows[0]["Start_Timestamp"]), None
tot = 0
for r in rows:
	s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
	n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
	n = re.sub(r"^void ", "", n)[:58]
	gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
	print(f"{(s - t0) / 1e3:9.1f} us  {n:58s} grid {r['Grid_Size_X']:>8s} wg {r['Workgroup_Size_X']:>4s}  {(e - s) / 1e3:7.2f} us  gap {gap:6.2f}")
	prev_end = e
	tot += e - s
print(f"kernels {tot / 1e3:.1f} us, wall {(prev_end - t0) / 1e3:.1f} us")
