"""GPU idle time inside the timed train steps, from a rocprofv3 --kernel-trace CSV (not a test):
   python tests/probe/trace_gaps.py <kernel_trace.csv> [steps]
Union of the kernel intervals over all queues against the wall span of the last `steps` optimizer steps (adamw_ema_kernel marks a step's end);
gaps are classified by the kernel that ends before them."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
ends = [e for s, e, n in ev if "adamw_ema_kernel" in n]
assert len(ends) > steps, len(ends)
t0, t1 = ends[-steps - 1], ends[-1]
win = [(s, e, n) for s, e, n in ev if s >= t0 and e <= t1]
busy, cur_s, cur_e, last_name = 0, None, None, None
gaps = defaultdict(lambda: [0, 0])
hist = defaultdict(int)
for s, e, n in win:
    if cur_e is None:
        cur_s, cur_e, last_name = s, e, n
        continue
    if s > cur_e:
        busy += cur_e - cur_s
        g = s - cur_e
        gaps[last_name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]][0] += g
        gaps[last_name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]][1] += 1
        hist[min(int(g / 1000), 20)] += 1
        cur_s, cur_e, last_name = s, e, n
    elif e > cur_e:
        cur_e, last_name = e, n
busy += cur_e - cur_s
wall = t1 - t0
print(f"steps {steps}: wall {wall / steps / 1e6:.3f} ms/step, busy {busy / steps / 1e6:.3f}, idle {(wall - busy) / steps / 1e6:.3f} ms/step, "
      f"{len(win) / steps:.0f} launches/step, sum of kernel durations {sum(e - s for s, e, n in win) / steps / 1e6:.3f} ms/step")
print("gap histogram (us: count per step):", {k: round(v / steps, 1) for k, v in sorted(hist.items())})
for name, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f"  after {name:60s} {t / steps / 1e3:8.1f} us/step in {c / steps:6.1f} gaps")
