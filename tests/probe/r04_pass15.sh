set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4q
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib/exp
for i in 1 2; do
  echo "== non-temporal (product)"; python tests/perf_wgrad43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/' | head -4
  echo "== plain"; VDIFF_HIP_LIB=$L/libvd_nt0.so python tests/perf_wgrad43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/' | head -4
done > $OUT/nt.txt 2>&1
grep -v "^+" $OUT/nt.txt | cut -c1-170
cat > /tmp/opt_time.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "v-diffusion-torch_amd"))
from v_diffusion import _hip as H
n = 60_806_404
p, g, m, v, e = [torch.randn(n, device="cuda") for _ in range(5)]
v.abs_()
gn = torch.ones(1, device="cuda")
f = lambda: H.adamw_ema(p, g, m, v, e, gn, 1.0, 2e-4, 0.9, 0.999, 1e-8, 0.001, 0.1, 0.001, 0.9999)
for _ in range(3): f()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): f()
b.record(); torch.cuda.synchronize()
t = a.elapsed_time(b) / 20
print(f"adamw_ema {t * 1e3:.1f} us  {9 * 4.0 * n / t / 1e9:.2f} TB/s")
PY
for i in 1 2; do echo "nt:"; python /tmp/opt_time.py; echo "plain:"; VDIFF_HIP_LIB=$L/libvd_nt0.so python /tmp/opt_time.py; done 2>&1 | grep -v amdgpu.ids
