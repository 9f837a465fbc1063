#!/usr/bin/env python3
"""Turn the CSVs that tools/collect_profile.sh left under gpurun_out/profile_<tag>/ into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats summary (per kernel: calls, total, average)
  profiles/<tag>_kernel_by_shape.csv   the same trace grouped by (kernel, grid) = per GEMM shape
  profiles/<tag>_hbm_traffic.csv       FETCH_SIZE / WRITE_SIZE per (kernel, grid), corrected as MI355X_MICROARCH.md prescribes
  profiles/<tag>_train_step_breakdown.csv   per (kernel, grid) time inside the timed optimizer steps
  profiles/<tag>_train_step_by_dispatch.csv the same window split by DISPATCH ORDER: launch k of a kernel inside a step is the same problem in every step, so the
                                        persistent-grid kernels (always 256 workgroups) come apart into one row per launch slot (logits GEMM, each layer's dW ...)
  profiles/<tag>_mfma_util.csv         SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE per (kernel, grid)
  profiles/roofline_traffic.json       HBM bytes per launch of the kernels bench.py prices (read by bench.py for roofline.traffic)
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
	return [float(r["Counter_Value"]) for r in rows if re.search(r"gemm256p?_kernel<0", r["Kernel_Name"])]


fd, wd = per_dispatch("fetch"), per_dispatch("write")
# (its output is rows x 6912 bf16 with rows = the non-padded output positions of the step's batch: the largest writes of any gemm256 dispatch)
wmax = max(wd[:min(len(fd), len(wd))] or [0.0])
sel = [i for i in range(min(len(fd), len(wd))) if wd[i] > 0.8 * wmax and wd[i] * 1024 > 0.3 * 57344 * 6912 * 2]
dom_val = int(sum(2 * fd[i] * 1024 + wd[i] * 1024 for i in sel) / len(sel)) if sel else None
dom = {"logits": dom_val}
# the dominant kernel by time share: wgrad256_kernel<8> on the in-projection gradient.  Its launches share the 256-workgroup grid with the out-projection
# gradient (a quarter of the operand bytes): told apart per dispatch by what they fetch; the fixed-order reduction that follows each is added.
def per_dispatch_named(kind, needle, grid=None):
	rows = csv.DictReader(open(newest(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0]))
	return [float(r["Counter_Value"]) for r in rows if needle in r["Kernel_Name"] and (grid is None or r["Grid_Size"] == grid)]


def commonest_grid(kind, needle):
	rows = csv.DictReader(open(newest(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0]))
	c = collections.Counter(r["Grid_Size"] for r in rows if needle in r["Kernel_Name"])
	return c.most_common(1)[0][0] if c else None


# (the layer gradients' launches: the grid with the most dispatches -- six per step against one for the logits layer)
WG8 = "wgrad256p_kernel<8>" if per_dispatch_named("fetch", "wgrad256p_kernel<8>") else "wgrad256_kernel<8>"
wf, ww = per_dispatch_named("fetch", WG8, "131072"), per_dispatch_named("write", WG8, "131072")
rgrid = commonest_grid("fetch", "wgrad_reduce_kernel<8>")
rf, rw = per_dispatch_named("fetch", "wgrad_reduce_kernel<8>", rgrid), per_dispatch_named("write", "wgrad_reduce_kernel<8>", rgrid)
wg_val = None
if wf and ww:
	n = min(len(wf), len(ww))
	big = [i for i in range(n) if wf[i] > 0.75 * max(wf[:n])]
	wg_val = int(sum(2 * wf[i] * 1024 + ww[i] * 1024 for i in big) / len(big))
	if rf and rw:
		m = min(len(rf), len(rw))
		wg_val += int(sum(2 * rf[i] * 1024 + rw[i] * 1024 for i in range(m)) / m)
# the largest class of the step: every dispatch of the bf16-store 256 x 256 tile kernel (QKV x layers, in-projection dX x layers, logits, logits dX, prefix MLP), per
# optimizer step (the PMC passes run 2 warm-up + 5 timed steps of the training-only bench)
def class_per_step(kind):
	"""(sum of the counter over the class's dispatches inside optimizer steps, steps): a step ends with its adamw_kernel dispatch; what follows the last one (bench.py's
	isolated logits launches) is left out."""
	rows = list(csv.DictReader(open(newest(os.path.join(src, kind, "*", "*_counter_collection.csv"))[0])))
	last = max((i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]), default=-1)
	steps = sum(1 for r in rows[:last + 1] if "adamw_kernel" in r["Kernel_Name"])
	vals = [float(r["Counter_Value"]) for r in rows[:last + 1] if re.search(r"gemm256p?_kernel<0", r["Kernel_Name"])]
	return vals, steps


cf, N_STEPS = class_per_step("fetch")
cw, _ = class_per_step("write")
ncls = min(len(cf), len(cw))
cls_val = int((2 * sum(cf[:ncls]) * 1024 + sum(cw[:ncls]) * 1024) / N_STEPS) if ncls and N_STEPS else None
json.dump({"tag": tag, "kernel": "gemm256p_kernel<0> (STORE_BF16) logits GEMM [non-padded rows of 57344 x 6912 x 512]", "hbm_bytes_per_launch": (list(dom.values()) or [None])[0],
           "gemm256_class_hbm_bytes_per_step": cls_val, "gemm256_class_dispatches_per_step": round(ncls / max(N_STEPS, 1), 2),
           "wgrad_kernel": WG8 + " + wgrad_reduce_kernel<8>, a layer's in-projection (+ out-projection, when paired) gradient launch, K = packed rows", "wgrad_in_proj_hbm_bytes_per_launch": wg_val,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, KB units, FETCH doubled (gfx950 correction)",
           # the tree the passes ran on (written by tools/collect_profile.sh ON the GPU box): bench.py reports these figures only for the same tree
           "source_sha16": (open(os.path.join(src, "source_fingerprint.txt")).read().strip() if os.path.exists(os.path.join(src, "source_fingerprint.txt")) else None)},
          open(os.path.join(dst, "roofline_traffic.json"), "w"), indent=1)

# MFMA utilisation pass
mf = newest(os.path.join(src, "mfma", "*", "*_counter_collection.csv"))
if mf:
	acc = collections.OrderedDict()
	for r in csv.DictReader(open(mf[0])):
		key = (clean(r["Kernel_Name"]), r["Grid_Size"])
		a = acc.setdefault(key, collections.defaultdict(float))
		a[r["Counter_Name"]] += float(r["Counter_Value"])
		a["_n_" + r["Counter_Name"]] += 1
	with open(os.path.join(dst, f"{tag}_mfma_util.csv"), "w", newline="") as f:
		w = csv.writer(f)
		# units (checked against the kernels whose MFMA count per launch is known): SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD, SQ_BUSY_CU_CYCLES the
		# busy cycles of every CU, GRBM_GUI_ACTIVE comes back summed over the 8 XCDs -- so the MFMA pipe utilisation of a SIMD, while its CU is busy, is
		# MFMA_BUSY / (4 * BUSY_CU), and over the launch's wall time MFMA_BUSY / (GUI / 8 * 256 CUs * 4 SIMDs)
		w.writerow(["kernel", "grid", "launches", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE_sum_of_8_XCDs", "mfma_util_per_simd_while_cu_busy",
		            "mfma_util_per_simd_over_wall_cycles"])
		for key, a in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)):
			mb, bc, gui = a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), a.get("SQ_BUSY_CU_CYCLES", 0.0), a.get("GRBM_GUI_ACTIVE", 0.0)
			n = int(a.get("_n_SQ_VALU_MFMA_BUSY_CYCLES", 0))
			w.writerow([key[0], key[1], n, int(mb), int(bc), int(gui), round(mb / (4 * bc), 4) if bc else "", round(mb / (gui / 8 * 256 * 4), 4) if gui else ""])
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
	# (bench.py --steps 5 --warmup 2 --repeats 1: launches 3..7 of adamw_kernel close the timed steps; the five behind them are the event-instrumented pass)
	opt_ends = opt_ends[:2 + nsteps] if len(opt_ends) >= 2 + 2 * nsteps else opt_ends
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
	# by dispatch order: the k-th launch of a kernel inside each step window
	step_of = lambda r: next(i for i in range(nsteps) if int(r["End_Timestamp"]) <= opt_ends[-nsteps + i])
	slots = collections.OrderedDict()
	counters_in_step = collections.defaultdict(int)
	for r in rows:
		if not (lo < int(r["End_Timestamp"]) <= hi):
			continue
		name = clean(r["Kernel_Name"])
		st_i = step_of(r)
		k = counters_in_step[(st_i, name, r["Grid_Size_X"])]
		counters_in_step[(st_i, name, r["Grid_Size_X"])] += 1
		a = slots.setdefault((name, r["Grid_Size_X"], k), [0, 0])
		a[0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
		a[1] += 1
	with open(os.path.join(dst, f"{tag}_train_step_by_dispatch.csv"), "w", newline="") as f:
		w = csv.writer(f)
		w.writerow(["kernel", "grid_x", "launch_slot_in_step", "steps_seen", "avg_us"])
		for (n, gx, k), (t, c) in slots.items():
			if any(tagname in n for tagname in ("gemm256_kernel", "gemm256p_kernel", "wgrad256_kernel", "wgrad256p_kernel", "wgrad_reduce_kernel", "gemm_kernel", "ffn_", "skinny_")):
				w.writerow([n, gx, k, c, round(t / c / 1e3, 2)])
# HBM bytes per optimizer step: launches per step (train-only trace window) x bytes per launch (PMC passes)
bd = os.path.join(dst, f"{tag}_train_step_breakdown.csv")
if os.path.exists(bd) and traffic:
	rows_out, total = [], 0.0
	for r in csv.DictReader(open(bd)):
		if not r["grid_x"]:
			continue
		b = traffic.get(f"{r['kernel']}|{r['grid_x']}")
		if b is None:
			continue
		per_step = b * float(r["launches_per_step"])
		total += per_step
		rows_out.append((per_step, r["kernel"], r["grid_x"], r["launches_per_step"], b, r["us_per_step"]))
	with open(os.path.join(dst, f"{tag}_hbm_per_step.csv"), "w", newline="") as f:
		w = csv.writer(f)
		w.writerow(["kernel", "grid_x", "launches_per_step", "hbm_bytes_per_launch", "hbm_GB_per_step", "us_per_step", "TB_per_s"])
		for per_step, n, gx, lps, b, us in sorted(rows_out, reverse=True):
			w.writerow([n, gx, lps, b, round(per_step / 1e9, 3), us, round(per_step / float(us) / 1e6, 2) if float(us) > 0 else ""])
		w.writerow(["TOTAL HBM bytes per optimizer step (kernels with counters)", "", "", "", round(total / 1e9, 2), "", ""])
	try:  # bench.py reports it next to the MFMA fraction of the whole step
		rt = json.load(open(os.path.join(dst, "roofline_traffic.json")))
		rt["train_step_hbm_bytes"] = int(total)
		json.dump(rt, open(os.path.join(dst, "roofline_traffic.json"), "w"), indent=1)
	except (OSError, ValueError):
		pass
bl = os.path.join(src, "bench_line_under_profiler.json")
if os.path.exists(bl):
	shutil.copy(bl, os.path.join(dst, f"{tag}_bench_line_under_profiler.json"))
print("wrote", sorted(os.listdir(dst)))
