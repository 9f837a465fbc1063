"""The counted `vmcnt` waits of wgrad256p_kernel's K loop (csrc/wgrad.hip), checked by COUNT for every part length -- not by timing.

Round 5's perturbation builds showed that timing cannot exercise these waits at all: with six half-tiles of staging distance the data has always landed, wait or no
wait ("Round 5").  A wrong immediate in the TAIL of a part (the last two K-tiles take their counts from the K-tiles left) would therefore pass every
GPU test until the day the memory system is slow.  Under data parallel with a CU budget the part count -- and with it the part length, odd or even, long or a single
K-tile -- varies, so this test walks the schedule for EVERY length 1 .. 40 and both tile heights:

  * the wait expressions, the stage conditions and the ladder of available immediates are READ OUT OF THE SOURCE (a change there changes what is checked here);
  * a wave's vector-memory operations retire in issue order, so after `s_waitcnt vmcnt(n)` everything but its n youngest operations has landed;
  * RAW: a half-tile read in the LOAD segment of phase s must have been retired by a wait of an EARLIER phase (every wave passes a barrier between that wait and the read:
    the one-phase-behind rule of the staggered wave groups);
  * WAR: the LDS region of a half-tile is restaged no earlier than two phases after its last read;
  * nothing is waited for that was never issued (a count larger than what is outstanding is merely useless; a wait on a half-tile that does not exist would be a bug in
    the model of the loop), and every K-tile of the part is read exactly once.
"""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "novic_amd", "csrc", "wgrad.hip")).read()


def c_to_py(expr: str) -> str:
	"""A C conditional expression -> Python (`a ? b : c` -> `(b if a else c)`), recursively; everything else passes through."""
	expr = expr.strip()
	depth = 0
	for i, ch in enumerate(expr):
		if ch == "(":
			depth += 1
		elif ch == ")":
			depth -= 1
		elif ch == "?" and depth == 0:
			level, d2 = 0, 0
			for j in range(i + 1, len(expr)):
				c = expr[j]
				if c == "(":
					d2 += 1
				elif c == ")":
					d2 -= 1
				elif c == "?" and d2 == 0:
					level += 1
				elif c == ":" and d2 == 0:
					if level == 0:
						return f"(({c_to_py(expr[i + 1:j])}) if ({c_to_py(expr[:i])}) else ({c_to_py(expr[j + 1:])}))"
					level -= 1
			raise ValueError(expr)
	if expr.startswith("(") and expr.endswith(")") and _balanced(expr[1:-1]):
		return "(" + c_to_py(expr[1:-1]) + ")"
	out, i = "", 0
	while i < len(expr):  # parenthesised sub-expressions may hold conditionals of their own
		if expr[i] == "(":
			j, d = i, 0
			while True:
				d += expr[j] == "("
				d -= expr[j] == ")"
				if d == 0:
					break
				j += 1
			out += "(" + c_to_py(expr[i + 1:j]) + ")"
			i = j + 1
		else:
			out += expr[i]
			i += 1
	return out


def _balanced(s: str) -> bool:
	d = 0
	for ch in s:
		d += ch == "("
		d -= ch == ")"
		if d < 0:
			return False
	return d == 0


def _kernel_text() -> str:
	start = SRC.index("void wgrad256p_kernel(")
	return SRC[start:SRC.index("wgrad_reduce_kernel", start)]


def _schedule():
	k = _kernel_text()
	ladder = [int(x) for x in re.findall(r"vm_wait_imm<(\d+)>\(\);", SRC[SRC.index("void vm_wait_dyn("):SRC.index("void wgrad256p_kernel(")])]
	steady = re.search(r"STEADY_WAIT = WGRAD_DIAG >= 2 \? 63 : (.+?);", k).group(1)
	dyn = re.findall(r"vm_wait_dyn\((.+)\);", k)
	assert len(dyn) == 3, dyn  # phase 1, phase 3, prologue -- in source order
	stage_conds = re.findall(r"if \(STEADY \|\| (rem > \d)\) stage_half\((bf(?: \^ 1)?), kt \+ (\d), C(\d)\{\}\);", k)
	assert len(stage_conds) == 4, stage_conds
	loops = re.search(r"for \(; (kt \+ \d < ke); kt \+= 2\) \{  // both K-tiles of the trip stage K-tiles that exist", k).group(1)
	return dict(ladder=sorted(ladder, reverse=True), steady=c_to_py(steady), phase1=c_to_py(dyn[0]), phase3=c_to_py(dyn[1]), prologue=c_to_py(dyn[2]), stages=stage_conds, steady_loop=loops)


S = _schedule()


def test_the_source_still_has_the_shape_this_model_reads():
	assert S["ladder"] == [8, 6, 5, 4, 3, 2, 0]
	assert [(c, b, int(d), int(q)) for c, b, d, q in S["stages"]] == [("rem > 1", "bf ^ 1", 1, 2), ("rem > 1", "bf ^ 1", 1, 3), ("rem > 2", "bf", 2, 0), ("rem > 2", "bf", 2, 1)]
	assert S["steady_loop"] == "kt + 3 < ke"
	assert eval(S["steady"], dict(NA=2)) == 8 and eval(S["steady"], dict(NA=1)) == 6


class Wave:
	"""One wave's vector-memory queue: LDS-DMA pieces in issue order, `wait(n)` retires all but the n youngest."""

	def __init__(self, NA):
		self.NA, self.issued, self.retired = NA, 0, 0
		self.end_of = {}      # (kt, q) -> issue count when its last piece was issued
		self.staged_at = {}   # (kt, q) -> global phase of the stage (-1: prologue)
		self.last_read = {}   # (buffer, q) -> (kt, global phase) of the last read of that LDS region
		self.reads = {}

	def stage(self, kt, q, phase, kb):
		region = ((kt - kb) & 1, q)
		prev = self.last_read.get(region)
		if prev is not None:
			assert prev[0] == kt - 2, f"restaging {region} with K-tile {kt} over K-tile {prev[0]}"
			assert phase - prev[1] >= 2, f"WAR: K-tile {kt} q{q} staged at phase {phase}, region last read at phase {prev[1]}"
		else:
			assert kt - kb < 2
		assert (kt, q) not in self.end_of
		self.issued += 2 if q % 2 == 0 else self.NA
		self.end_of[(kt, q)] = self.issued
		self.staged_at[(kt, q)] = phase

	def wait(self, n, ladder):
		imm = next(v for v in ladder if v <= n)  # vm_wait_dyn rounds DOWN to an available immediate
		self.retired = max(self.retired, self.issued - imm)
		return imm

	def read(self, kt, q, phase, kb, retired_before_phase):
		assert (kt, q) in self.end_of, f"phase {phase} reads K-tile {kt} q{q}, which was never staged"
		assert self.end_of[(kt, q)] <= retired_before_phase, (f"RAW: phase {phase} reads K-tile {kt} q{q} (issue count {self.end_of[(kt, q)]}), "
		                                                      f"but the waits of earlier phases only retired {retired_before_phase} operations")
		self.last_read[((kt - kb) & 1, q)] = (kt, phase)
		self.reads[(kt, q)] = self.reads.get((kt, q), 0) + 1


def walk(NA: int, nkt: int):
	kb, ke = 3, 3 + nkt  # (a part in the middle of the token range: only differences matter)
	w = Wave(NA)
	env = dict(NA=NA, kb=kb, ke=ke)
	for q in range(4):
		w.stage(kb, q, -1, kb)
	if kb + 1 < ke:
		w.stage(kb + 1, 0, -1, kb)
		w.stage(kb + 1, 1, -1, kb)
	w.wait(eval(S["prologue"], env), S["ladder"])
	waits = []

	def ktile(kt, steady):
		rem = ke - kt
		e = dict(env, rem=rem)
		P = 4 * (kt - kb)
		conds = [steady or eval(c, e) for c, _, _, _ in S["stages"]]
		# phase 0
		r0 = w.retired
		w.read(kt, 0, P, kb, r0); w.read(kt, 1, P, kb, r0)
		if conds[0]: w.stage(kt + 1, 2, P, kb)
		# phase 1
		r1 = w.retired
		w.read(kt, 1, P + 1, kb, r1)
		if conds[1]: w.stage(kt + 1, 3, P + 1, kb)
		waits.append(w.wait(eval(S["steady"], e) if steady else eval(S["phase1"], e), S["ladder"]))
		# phase 2
		r2 = w.retired
		w.read(kt, 2, P + 2, kb, r2); w.read(kt, 3, P + 2, kb, r2)
		if conds[2]: w.stage(kt + 2, 0, P + 2, kb)
		# phase 3
		r3 = w.retired
		w.read(kt, 3, P + 3, kb, r3)
		if conds[3]: w.stage(kt + 2, 1, P + 3, kb)
		waits.append(w.wait(eval(S["steady"], e) if steady else eval(S["phase3"], e), S["ladder"]))

	kt = kb
	while eval(S["steady_loop"], dict(kt=kt, ke=ke)):
		ktile(kt, True); ktile(kt + 1, True)
		kt += 2
	while kt < ke:
		ktile(kt, False)
		if kt + 1 < ke:
			ktile(kt + 1, False)
		kt += 2
	return w, waits


@pytest.mark.parametrize("NA", [2, 1], ids=["256-row tiles", "128-row tiles"])
def test_every_part_length_retires_what_it_reads(NA):
	for nkt in range(1, 41):
		w, waits = walk(NA, nkt)
		kb = 3
		# every half-tile of every K-tile of the part was staged once and read (q 0 and 2: once, by phase 0 / 2; q 1 and 3: by two phases)
		assert sorted(w.end_of) == [(kb + t, q) for t in range(nkt) for q in range(4)], nkt
		assert all(w.reads[(kb + t, q)] == (1 if q % 2 == 0 else 2) for t in range(nkt) for q in range(4)), nkt
		# nothing is left in flight behind the last K-tile's last wait that a later read would need -- and the last wait of a part drains the queue (the partial sums' stores follow)
		assert waits[-1] == 0 and w.retired == w.issued, nkt
		# the steady trips keep the full staging distance in flight (a tail wait never, the prologue aside): the loop is not draining where it need not
		steady_trips = max(0, (nkt - 2) // 2) * 2
		assert all(v == 2 * (2 + NA) for v in waits[:2 * steady_trips]), nkt


def test_a_wrong_tail_count_is_caught():
	"""The model has teeth: one more operation left in flight at the second-to-last K-tile's phase-3 wait is a read of a half-tile that has not landed."""
	good = S["phase3"]
	try:
		S["phase3"] = good.replace("(2 + NA)", "(3 + NA)", 2)
		with pytest.raises(AssertionError, match="RAW"):
			for nkt in range(1, 9):
				walk(2, nkt)
	finally:
		S["phase3"] = good
