#!/usr/bin/env python3
"""A rocprofv3 kernel trace read per STREAM: busy time, span, and -- for the stream that carries the ViT tower (the one with im2col launches) -- every forward's duration and the
idle time between two forwards; the same window's decode streams beside it.  python3 tools/streams_of_trace.py DIR"""
import collections
import csv
import glob
import os
import sys

fs = sorted(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv"), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(fs[-1])), key=lambda r: int(r["Start_Timestamp"]))
by = collections.defaultdict(list)
for r in rows:
	by[(r["Queue_Id"], r["Stream_Id"])].append(r)
t_end = int(rows[-1]["End_Timestamp"])
win0 = t_end - (t_end - int(rows[0]["Start_Timestamp"])) // 3  # the last third: steady state of the timed run
print(f"{len(rows)} launches on {len(by)} (queue, stream) pairs; window = the last third of the trace ({(t_end - win0) / 1e6:.1f} ms)")
for key, rs in sorted(by.items(), key=lambda kv: -len(kv[1])):
	w = [r for r in rs if int(r["Start_Timestamp"]) >= win0]
	if not w:
		continue
	busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w)
	span = int(w[-1]["End_Timestamp"]) - int(w[0]["Start_Timestamp"])
	names = collections.Counter(r["Kernel_Name"].split("(")[0][-40:] for r in w).most_common(2)
	print(f"queue {key[0]} stream {key[1]}: {len(w):6d} launches, busy {busy / 1e6:8.2f} ms of a span of {span / 1e6:8.2f} ms ({100 * busy / max(span, 1):5.1f} %)  e.g. {names[0][0]}")
	starts = [i for i, r in enumerate(w) if "im2col" in r["Kernel_Name"]]
	if len(starts) >= 3:
		print("    tower forwards in the window (duration from im2col to the last launch before the next im2col; idle = from there to the next im2col):")
		for a, b in zip(starts[:-1], starts[1:]):
			t0, t1, t2 = int(w[a]["Start_Timestamp"]), int(w[b - 1]["End_Timestamp"]), int(w[b]["Start_Timestamp"])
			ksum = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in w[a:b])
			print(f"      forward {(t1 - t0) / 1e3:9.1f} us (kernels {ksum / 1e3:9.1f}), then idle {(t2 - t1) / 1e3:8.1f} us")
