#!/usr/bin/env python3
"""Turn the CSVs that tools/collect_profile.sh left under gpurun_out/profile_<tag>/ into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (per kernel: calls, total, average)
  profiles/<tag>_kernel_by_shape.csv   the same trace grouped by (kernel, grid) = per GEMM shape
  profiles/<tag>_hbm_traffic.csv       FETCH_SIZE / WRITE_SIZE per (kernel, grid), corrected as MI355X_MICROARCH.md prescribes
  profiles/roofline_traffic.json       HBM bytes per launch of the dominant kernel (read by bench.py for roofline.traffic)
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
clean = lambda n: re.sub(r"\(anonymous namespace\)::", "", n)
_glob = glob.glob


def newest(pattern):
	"""gpurun merges every run's files into the same directory: take the most recent match."""
	m = sorted(_glob(pattern), key=os.path.getmtime)
	return m[-1:]



stats = newest(os.path.join(src, "trace", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))

trace = list(csv.DictReader(open(newest(os.path.join(src, "trace", "*", "*_kernel_trace.csv"))[0])))
agg = collections.OrderedDict()
for r in trace:
	key = (clean(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Z"])
	a = agg.setdefault(key, [0, 0])
	a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
	a[1] += 1
with open(os.path.join(dst, f"{tag}_kernel_by_shape.csv"), "w", newline="") as f:
	w = csv.writer(f)
	w.writerow(["kernel", "grid_x", "grid_z", "calls", "total_us", "avg_us"])
	for (n, gx, gz), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
		w.writerow([n, gx, gz, c, round(t / 1e3, 1), round(t / c / 1e3, 2)])


def counters(kind):
	rows = list(csv.DictReader(open(newest(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0])))
	out = collections.OrderedDict()
	for r in rows:
		key = (clean(r["Kernel_Name"]), r["Grid_Size"])
		a = out.setdefault(key, [0.0, 0])
		a[0] += float(r["Counter_Value"])
		a[1] += 1
	return out


fetch, write = counters("fetch"), counters("write")
traffic = {}
with open(os.path.join(dst, f"{tag}_hbm_traffic.csv"), "w", newline="") as f:
	w = csv.writer(f)
	w.writerow(["kernel", "grid", "launches", "FETCH_SIZE_KB_raw_per_launch", "fetch_bytes_corrected_x2", "WRITE_SIZE_KB_per_launch", "write_bytes", "hbm_bytes_per_launch"])
	for key in sorted(fetch, key=lambda k: -(fetch[k][0])):
		fr, n = fetch[key]
		wr, wn = write.get(key, (0.0, 1))
		fb = 2 * fr / n * 1024  # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)
		wb = wr / max(wn, 1) * 1024
		w.writerow([key[0], key[1], n, round(fr / n, 1), int(fb), round(wr / max(wn, 1), 1), int(wb), int(fb + wb)])
		traffic[f"{key[0]}|{key[1]}"] = int(fb + wb)
# dominant kernel launch = the logits GEMM.  The persistent 256^2 kernel always launches 256 workgroups, so its launches are told apart per
# dispatch (same command in both passes => same dispatch order): the logits launches are the gemm256 dispatches that WRITE 57344*6912*2 bytes.
def per_dispatch(kind):
	rows = csv.DictReader(open(newest(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0]))
	return [float(r["Counter_Value"]) for r in rows if "gemm256_kernel" in r["Kernel_Name"]]


fd, wd = per_dispatch("fetch"), per_dispatch("write")
# (its output is rows x 6912 bf16 with rows = the non-padded output positions of the step's batch: the largest writes of any gemm256 dispatch)
wmax = max(wd[:min(len(fd), len(wd))] or [0.0])
sel = [i for i in range(min(len(fd), len(wd))) if wd[i] > 0.8 * wmax and wd[i] * 1024 > 0.3 * 57344 * 6912 * 2]
dom_val = int(sum(2 * fd[i] * 1024 + wd[i] * 1024 for i in sel) / len(sel)) if sel else None
dom = {"logits": dom_val}
json.dump({"tag": tag, "kernel": "gemm256_kernel<0, 4> (STORE_BF16) logits GEMM [non-padded rows of 57344 x 6912 x 512]", "hbm_bytes_per_launch": (list(dom.values()) or [None])[0],
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KB units, FETCH doubled (gfx950 correction)"},
          open(os.path.join(dst, "roofline_traffic.json"), "w"), indent=1)
# train-only pass: the 5 timed optimizer steps = everything between the end of the 2nd (last warm-up) and the end of the 7th adamw_kernel launch
# (the isolated dominant-kernel timing bench.py does afterwards falls outside that window)
tr = newest(os.path.join(src, "train", "*", "*_kernel_trace.csv"))
if tr:
	st = newest(os.path.join(src, "train", "*", "*_kernel_stats.csv"))
	if st:
		shutil.copy(st[0], os.path.join(dst, f"{tag}_train_only_kernel_stats.csv"))
	rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
	opt_ends = [int(r["End_Timestamp"]) for r in rows if "adamw_kernel" in r["Kernel_Name"]]
	nsteps = 5
	lo, hi = opt_ends[-nsteps - 1], opt_ends[-1]
	agg = collections.OrderedDict()
	for r in rows:
		if not (lo < int(r["End_Timestamp"]) <= hi):
			continue
		key = (clean(r["Kernel_Name"]), r["Grid_Size_X"], r["Grid_Size_Z"])
		a = agg.setdefault(key, [0, 0])
		a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
		a[1] += 1
	with open(os.path.join(dst, f"{tag}_train_step_breakdown.csv"), "w", newline="") as f:
		w = csv.writer(f)
		w.writerow(["kernel", "grid_x", "grid_z", "launches_per_step", "us_per_step", "avg_us"])
		tot = 0.0
		for (n, gx, gz), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
			w.writerow([n, gx, gz, round(c / nsteps, 2), round(t / nsteps / 1e3, 1), round(t / c / 1e3, 2)])
			tot += t / nsteps / 1e3
		w.writerow(["TOTAL kernel time per optimizer step", "", "", "", round(tot, 1), ""])
		w.writerow(["wall time per optimizer step (window / 5)", "", "", "", round((hi - lo) / nsteps / 1e3, 1), ""])
bl = os.path.join(src, "bench_line_under_profiler.json")
if os.path.exists(bl):
	shutil.copy(bl, os.path.join(dst, f"{tag}_bench_line_under_profiler.json"))
print("wrote", sorted(os.listdir(dst)))
