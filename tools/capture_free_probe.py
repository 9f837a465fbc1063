"""Which HIP objects may ANOTHER host thread free while a hipGraph capture is open on this one?  (run on the GPU box: python tools/capture_free_probe.py)

Round 6: the review of round 5 asked for the capture abort to be excluded by construction and proposed torch's `capture_error_mode="thread_local"`.  The first test of that --
a second thread dropping its last references to a pinned buffer, a device tensor, an event and a captured graph while a capture was open -- ABORTED the process (no HIP error
text), so the question is asked here one object kind and one capture mode at a time, each combination in a process of its own (an abort takes the interpreter with it):
the table this prints is the evidence `ops.graph_capture` / `ops.hip_free_guard` are designed from ("Round 6")."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import gc, sys, threading, time, torch
sys.path.insert(0, %(root)r)
kind, mode = sys.argv[1], sys.argv[2]
dev = torch.device("cuda", 0)
x = torch.zeros(8, device=dev)
side = torch.cuda.Stream()
def make(kind):
    if kind == "pinned": return torch.empty(1 << 20).pin_memory()
    if kind == "pinned_used":
        t = torch.empty(1 << 20).pin_memory(); d = torch.empty(1 << 20, device=dev); d.copy_(t, non_blocking=True); torch.cuda.synchronize(); return t
    if kind == "device": return torch.empty(1 << 20, device=dev)
    if kind == "device_big": return torch.empty(1 << 28, device=dev)
    if kind == "event":
        e = torch.cuda.Event(); e.record(); return e
    if kind == "event_timing":
        e = torch.cuda.Event(enable_timing=True); e.record(); return e
    if kind == "stream": return torch.cuda.Stream()
    if kind in ("graph", "graph_pool"):
        g = torch.cuda.CUDAGraph()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.cuda.graph(g, stream=side):
            if kind == "graph_pool":
                y = x + 1  # an allocation inside the capture: the graph owns a private memory pool
            else:
                x.add_(0)
        torch.cuda.current_stream().wait_stream(side)
        return g
    if kind == "pin_alloc": return None
    if kind == "sync": return None
    raise SystemExit("unknown kind " + kind)
victim = [make(kind)]
torch.cuda.synchronize()
opened, done = threading.Event(), threading.Event()
err = []
def other():
    try:
        opened.wait(30)
        if kind == "pin_alloc": torch.empty(1 << 18).pin_memory()
        elif kind == "sync": torch.cuda.synchronize()
        else:
            victim.clear(); gc.collect()
    except BaseException as e:
        err.append(repr(e)[:200])
    finally:
        done.set()
t = threading.Thread(target=other); t.start()
g = torch.cuda.CUDAGraph()
side.wait_stream(torch.cuda.current_stream())
ok = True
try:
    with torch.cuda.stream(side), torch.cuda.graph(g, stream=side, capture_error_mode=mode):
        x.add_(1)
        opened.set()
        done.wait(30)
        x.add_(1)
except BaseException as e:
    ok = False
    err.append("capture: " + repr(e)[:200])
t.join()
torch.cuda.current_stream().wait_stream(side)
if ok:
    g.replay(); torch.cuda.synchronize()
    ok = float(x[0]) == 2.0
print("RESULT", "ok" if ok and not err else "error " + " | ".join(err))
'''


def main():
	kinds = ["pinned", "pinned_used", "device", "device_big", "event", "event_timing", "stream", "graph", "graph_pool", "pin_alloc", "sync"]
	modes = ["global", "thread_local", "relaxed"]
	code = CHILD % dict(root=ROOT)
	print(f"{'object freed / call made on the other thread':46s}" + "".join(f"{m:>16s}" for m in modes))
	for k in kinds:
		row = []
		for m in modes:
			r = subprocess.run([sys.executable, "-c", code, k, m], capture_output=True, text=True, timeout=120)
			line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT")]
			if r.returncode != 0 and not line:
				row.append(f"ABORT rc {r.returncode}")
			else:
				row.append(line[-1][7:][:60] if line else f"rc {r.returncode}")
		print(f"{k:46s}" + "".join(f"{v[:15]:>16s}" for v in row))
		for m, v in zip(modes, row):
			if not v.startswith("ok"):
				print(f"    {m}: {v}")


if __name__ == "__main__":
	main()
