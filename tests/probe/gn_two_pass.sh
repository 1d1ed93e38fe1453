# per-kernel times of the two-pass GroupNorm backward (reduction pass / apply pass) at the benchmark shapes (not a test)
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-gn2p}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VD_GN_TWO_PASS=1
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/tests/perf_gn.py > $OUT/out.txt 2>&1
cd $GRAFT_REPO_ROOT
grep -v amdgpu $OUT/out.txt
python3 - $OUT <<'PY'
import csv, glob, sys, collections
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].split("(")[0][-44:]
    if "gn_bwd" in n or "chan_reduce" in n or "gn_apply_kernel" in n:
        agg.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in agg.items():
    print(n, len(v), [round(x, 1) for x in v[5::21][:8]])
PY
find $OUT -name "*.csv" -delete
