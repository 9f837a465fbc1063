#!/usr/bin/env python3
"""Build-time audit of the hand-written counted `s_waitcnt vmcnt(N)` (N > 0) in the HIP sources.

A counted wait is sound on gfx950 only under the in-order rule (MI355X_MICROARCH.md: loads, stores, atomics and LDS-DMA retire in issue order; flat_*
excepted) and only if the N operations it leaves in flight are really the N youngest.  For every kernel that holds such a wait (found by the
;;#ASMSTART markers hipcc puts around inline asm) the audit checks in the generated gfx950 ISA that

  1. the kernel issues no flat_* memory instruction (those retire out of order: a counted wait says nothing about them),
  2. the kernel has no scratch_* instruction (a spilled kernel reloads through vector memory: extra, compiler-placed operations in the count),
  3. a wait that follows its requests in straight-line code (the staging waits of skinny.hip) is preceded, inside its basic block, by at least N
     vector-memory issues with no label in between -- i.e. the N youngest are the requests the comment says they are, on every path,
  4. for the waits whose N youngest were issued in the PREVIOUS loop iteration (gemm256.hip: the epilogue's stores of the last output tile), which
     no straight-line scan can see, the kernel's ISA holds a straight-line run of exactly N stores with no branch inside it behind an LDS-DMA group
     (the stores an interior tile issues unconditionally); listed for the reader, with the run lengths found.

Usage: python tools/audit_vmcnt.py [file.hip ...]   (default: every csrc/*.hip that contains a counted wait).  Exit code 1 on a violation."""
import os
import re
import subprocess
import sys
import tempfile

# Waits whose N youngest operations were issued on the far side of a branch or a loop back-edge: no straight-line scan can prove them, the argument is
# written here (and beside the wait in the source) and the audit only checks premises 1 and 2 for their kernels.
CROSS_BLOCK = {
	("gemm256_kernel", 16): "the 16 (12, 8) youngest are the nontemporal stores store_tile issued for the previous INTERIOR output tile -- unconditional, no lane-dependent "
	                        "branch -- and store_tile returns 0 (-> vmcnt(0)) for every other path; the LDS-DMA of K-tile 1 was issued before those stores",
	("gemm256_kernel", 12): "as vmcnt(16): 256 x 192 tiles issue 12 stores",
	("gemm256_kernel", 8): "as vmcnt(16): 128-wide tiles issue 8 stores",
	("skinny_k128_resid_kernel", 16): "the 16 youngest are this iteration's 8 residual requests + 8 stores; a wave skips the stores only on the globally last tile, which has "
	                                  "no successor (has_next false -> this wait is not executed); conditional bias loads, if any, only add older operations",
}

# The 8-phase kernels (gemm256p_kernel, wgrad256p_kernel: cdna_hip_programming.md section 5): every counted wait leaves in flight the LDS-DMA pieces of the half-tiles
# staged BEHIND the half-tile(s) the next phase reads (+ the previous tile's epilogue stores, gemm256p only).  The count is a function of the schedule (phase s stages
# half-tile s + 6 of the stream -- gemm256p since round 4: phases 1 / 2 / 3 stage one / one / two half-tiles, phase 0 none; its four-phase form (256-row tiles): P0 one, P1 three; the waits sit one phase before the first read), written beside each wait in the source; waits outside the steady loop get their count
# from the number of K-tiles left (vm_wait_dyn rounds DOWN to an available immediate, which is always safe).  What the ISA can show, and audit_phased() checks: the
# steady loop issues no vector-memory operation other than LDS-DMA (nothing else can slip into the count), every LOAD segment's DMA pieces come as whole half-tile
# groups, and every counted wait of the loop leaves a whole number (1..5) of the most recent groups in flight.
PHASED = "256p_kernel"
PHASED_WHY = ("8-phase schedule: the N youngest operations are the LDS-DMA pieces of the half-tiles staged behind the one the next phase reads (+ the previous tile's "
              "epilogue stores); outside the steady loop N comes from the K-tiles left, rounded down (vm_wait_dyn)")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "novic_amd", "csrc")
VMEM = re.compile(r"^\s*(buffer_(load|store|atomic)\w*|global_(load|store|atomic)\w*)\b")
STORE = re.compile(r"^\s*(buffer_store\w*|global_store\w*)\b")
DMA = re.compile(r"^\s*buffer_load_dword\w*\s.*\blds\b")


def compile_asm(path):
	out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
	cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-ffp-contract=fast", "-S",
	       "--cuda-device-only", path, "-o", out]
	subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
	with open(out) as f:
		text = f.read()
	os.unlink(out)
	return text


def kernels(text):
	"""[(name, [lines])] for every function body in the assembly."""
	out, name, body = [], None, []
	for line in text.splitlines():
		m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", line)
		if m and not line.startswith(".L") and name is None:
			name, body = m.group(1), []
			continue
		if name is not None:
			if line.startswith(".Lfunc_end"):
				out.append((name, body))
				name = None
			else:
				body.append(line)
	return out


def audit_phased(name, body):
	"""Steady loops of an 8-phase kernel: [(loop label, DMA groups per trip, [(N, groups left in flight)]), ...], violations."""
	labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
	loops = []
	for i, l in enumerate(body):
		m = re.match(r"^\s*s_cbranch_\w+\s+(\.LBB\d+_\d+)", l)
		if m and m.group(1) in labels and labels[m.group(1)] < i:
			loops.append((labels[m.group(1)], i, m.group(1)))
	out, bad = [], 0
	for lo, hi, label in loops:
		seg = body[lo:hi]
		if not any("v_mfma" in l for l in seg) or any(re.match(r"^\.LBB", l) for l in seg[1:]):
			continue  # not an innermost straight-line loop body with matrix work
		events, in_asm = [], False
		for l in seg:
			if "#ASMSTART" in l:
				in_asm = True
			elif "#ASMEND" in l:
				in_asm = False
			elif DMA.match(l):
				events.append("D")
			elif VMEM.match(l):
				events.append("X")
			elif "s_barrier" in l:
				events.append("B")
			elif in_asm and (m := re.search(r"s_waitcnt\s+vmcnt\((\d+)\)", l)):
				events.append(int(m.group(1)))
		if not any(isinstance(e, int) and e > 0 for e in events):
			continue
		if "X" in events:
			bad += 1
		# DMA groups = runs of D between barriers (one stage_half per LOAD segment; gemm256p's phase 3 stages the two B half-tiles of a K-tile: a run of 4 = 2 + 2)
		split = lambda r: [2] * (r // 2) if (r > 2 and r % 2 == 0) else [r]  # (several half-tiles staged in one LOAD segment: phase 3 of the 8-phase loop 2 + 2, P1 of the four-phase loop 2 + 2 + 2)
		groups, run = [], 0
		for e in events:
			if e == "D":
				run += 1
			elif e == "B" and run:
				groups += split(run)
				run = 0
		if run:
			groups += split(run)
		waits = []
		pos_groups = []  # groups completed before each event, cyclically
		done = 0
		run = 0
		for e in events:
			if e == "D":
				run += 1
			elif isinstance(e, int) and e > 0:
				# the groups issued so far in this trip (the current LOAD segment's pieces precede its wait), then the previous trip's, youngest first
				recent = (split(run) if run else []) + list(reversed(groups[:done])) + list(reversed(groups))
				left, g = e, 0
				while left > 0 and g < len(recent):
					left -= recent[g]
					g += 1
				ok = left == 0 and 1 <= g <= 5
				waits.append((e, g if left == 0 else None))
				if not ok:
					bad += 1
			elif e == "B" and run:
				done += len(split(run))
				run = 0
		out.append((label, groups, waits))
	return out, bad


def audit(path):
	findings, bad = [], 0
	for name, body in kernels(compile_asm(path)):
		# the whole-line output stores of the 256-wide GEMM tiles must carry the non-temporal policy (streamed out past L2, so that the operand panels stay resident): a
		# source-level `if (knob) plain store; else nontemporal store;` was once merged by hipcc into ONE plain store and the hint silently lost (round 4)
		if re.search(r"gemm256p?_kernelILi0ELi[48]E", name):
			stores = [l for l in body if re.match(r"^\s*global_store_dwordx4\b", l)]
			nt = [l for l in stores if re.search(r"\bnt\b", l)]
			print(f"{os.path.basename(path)}: {name[:60]}: {len(nt)} of {len(stores)} global_store_dwordx4 are non-temporal{' VIOLATION' if len(nt) < 32 else ''}")
			if len(nt) < 32:
				bad += 1
		if PHASED in name:
			loops, b = audit_phased(name, body)
			bad += b
			if not loops:
				bad += 1
			for label, groups, waits in loops:
				print(f"{os.path.basename(path)}: {name[:60]}: steady loop {label}: LDS-DMA groups per trip {groups}, counted waits (N, half-tile groups left in flight) {waits}"
				      f"{' VIOLATION' if b else ''}")
		waits = []  # (index, N)
		in_asm = False
		for i, line in enumerate(body):
			if "#ASMSTART" in line:
				in_asm = True
			elif "#ASMEND" in line:
				in_asm = False
			elif in_asm:
				m = re.search(r"s_waitcnt\s+vmcnt\((\d+)\)", line)
				if m and int(m.group(1)) > 0:
					waits.append((i, int(m.group(1))))
		if not waits:
			continue
		flat = [l.strip() for l in body if re.match(r"^\s*flat_(load|store|atomic)", l)]
		scratch = [l.strip() for l in body if re.match(r"^\s*scratch_", l)]
		store_runs = []
		run, after_dma = 0, False
		for l in body:
			if DMA.match(l):
				after_dma, run = True, 0
			elif STORE.match(l):
				if after_dma:
					run += 1
			elif re.match(r"^\s*(s_cbranch|s_branch)", l) or re.match(r"^\.LBB", l) or (VMEM.match(l) and not STORE.match(l)):
				if after_dma and run:
					store_runs.append(run)
				after_dma, run = False, 0
		if PHASED in name:  # one summary line: the waits outside the steady loop (prologue, first K-tile behind an epilogue, stream tail) by count
			counts = sorted({n for _, n in waits})
			print(f"{os.path.basename(path)}: {name[:60]}: {len(waits)} counted waits in all, N in {counts} ({PHASED_WHY}); flat {len(flat)}, scratch {len(scratch)}")
			if flat or scratch:
				bad += 1
			continue
		for i, n in waits:
			k, j = 0, i - 1
			while j >= 0 and not re.match(r"^\.LBB", body[j]) and not re.match(r"^\s*(s_cbranch|s_branch)", body[j]):
				if VMEM.match(body[j]):
					k += 1
				j -= 1
			straight = k >= n
			argued = PHASED_WHY if PHASED in name else next((why for (kn, nn), why in CROSS_BLOCK.items() if kn in name and nn == n), None)
			findings.append(dict(kernel=name, n=n, straight_line_requests=k, straight=straight, flat=len(flat), scratch=len(scratch), store_runs=sorted(set(store_runs)), argued=argued))
			if flat or scratch:
				bad += 1
			if not straight and argued is None:
				bad += 1
	return findings, bad


def main(argv):
	files = argv or [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") and re.search(r'asm volatile\("s_waitcnt vmcnt\(([1-9]|%0)', open(os.path.join(CSRC, f)).read())]
	total_bad = 0
	for f in files:
		findings, bad = audit(f)
		total_bad += bad
		for r in findings:
			how = (f"straight-line: {r['straight_line_requests']} vector-memory issues precede it in its block" if r["straight"] else
			       f"cross-block ({'argued: ' + r['argued'] if r['argued'] else 'NOT ARGUED'}); store runs behind an LDS-DMA group in this kernel = {r['store_runs']}")
			print(f"{os.path.basename(f)}: {r['kernel'][:70]}: vmcnt({r['n']}): {how}; flat {r['flat']}, scratch {r['scratch']}")
	print("violations:", total_bad)
	return 1 if total_bad else 0


if __name__ == "__main__":
	sys.exit(main(sys.argv[1:]))
