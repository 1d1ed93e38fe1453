"""Same-box A/B of two builds of the kernel library (not a test): python tests/perf_ab.py <libA.so> <libB.so> ...
Each library is exercised in its own child process, alternating, three rounds; box-to-box variance on the pool is +-1.5 %,
so only numbers from one invocation are comparable."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys
sys.path[:0] = [%r, os.path.join(%r, "v-diffusion-torch_amd"), os.path.join(%r, "tests")]
from v_diffusion import _hip as H
H.LIB_PATH = sys.argv[1]
import torch
from perf_kernels import timeit
DEV = "cuda"
B = 128
out = []
for (Hh, Cin, Cout) in [(32, 256, 256), (32, 512, 256), (16, 256, 256)]:
    x = torch.randn(B, Hh, Hh, Cin, device=DEV); w = torch.randn(Cout, 9, Cin, device=DEV) * 0.02
    wd = torch.randn(Cin, 9, Cout, device=DEV) * 0.02; bias = torch.randn(Cout, device=DEV)
    y = torch.randn(B, Hh, Hh, Cout, device=DEV); dx = torch.empty(B, Hh, Hh, Cin, device=DEV); dw = torch.empty(Cout, Cin, 3, 3, device=DEV)
    part = torch.empty(H.stats_part_numel(B, Hh * Hh, Cout), device=DEV)
    fl = 2.0 * B * Hh * Hh * Cout * 9 * Cin
    for _ in range(2):   # first pass warms the clocks
        a = timeit(lambda: H.conv3x3(x, Cin, w, bias, y, Cout, B, Hh, Hh, Cin, Cout, stats_part=part), fl, "fwd+stats", iters=20)
        b = timeit(lambda: H.conv3x3(y, Cout, wd, None, dx, Cin, B, Hh, Hh, Cout, Cin), fl, "dgrad", iters=20)
        c = timeit(lambda: H.conv3x3_wgrad(x, Cin, y, Cout, B, Hh, Hh, Cin, Cout, dw, Cin, Cout), fl, "wgrad", iters=20)
    out.append((Hh, Cin, Cout, fl / a / 1e9, fl / b / 1e9, fl / c / 1e9))
print("RESULT", out)
''' % (ROOT, ROOT, ROOT)
libs = sys.argv[1:]
acc = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        r = subprocess.run([sys.executable, "-c", CHILD, l], capture_output=True, text=True, timeout=600)
        line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
        if not line:
            print(r.stdout[-2000:], r.stderr[-2000:]); sys.exit(1)
        acc[l].append(eval(line[0][7:]))
for l in libs:
    print(os.path.basename(l))
    for i in range(len(acc[l][0])):
        Hh, Cin, Cout = acc[l][0][i][:3]
        cols = [[run[i][k] for run in acc[l]] for k in (3, 4, 5)]
        print(f"  {Cin}->{Cout} @{Hh}: fwd+stats {max(cols[0]):6.1f}  dgrad {max(cols[1]):6.1f}  wgrad {max(cols[2]):6.1f}   (best of {len(acc[l])}; TFLOP/s)")
