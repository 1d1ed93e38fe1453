# Round-4 pass 4: same-box A/B of the V / dM layouts of the F(4x4,3x3) weight gradient, the whole GPU suite, the default bench line with the
# new fields, the 1-GPU slice of configs[4], RCCL stream priority x reserved CUs on one rank
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4d}
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib
for i in 1 2; do
  echo "== K-blocked (product)"; python tests/perf_wgrad43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/'
  echo "== plane-major"; VDIFF_HIP_LIB=$L/exp/libvd_planemajor.so python tests/perf_wgrad43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/'
done > $OUT/wgrad43_ab.txt 2>&1
B="python bench.py --steps 20 --warmup 5 --no-sample --no-cpu-baseline --no-secondary --no-extras"
for i in 1 2; do
  $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step K-blocked', j['ms_per_step'])"
  VDIFF_HIP_LIB=$L/exp/libvd_planemajor.so $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('step plane-major', j['ms_per_step'])"
done >> $OUT/wgrad43_ab.txt 2>&1
for hp in 0 1; do for n in 0 16; do
  echo "== reserve $n high-priority $hp"; VD_BENCH_FORCE_REDUCER=1 VD_RCCL_HIGH_PRIORITY=$hp VD_RESERVE_CUS=$n MASTER_PORT=$((29600 + n + hp)) $B 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(j['ms_per_step'], j['multi_gpu'])"
done; done > $OUT/dp_levers3.txt 2>&1
timeout 1200 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 900 python tests/probe/celeba_ddim250.py > $OUT/celeba_ddim250.json 2> $OUT/celeba_ddim250.err
timeout 2700 python -m pytest tests -q -m gpu 2>&1 | tail -30 > $OUT/gputests.txt
cat $OUT/wgrad43_ab.txt | grep -v "^+"; cat $OUT/dp_levers3.txt | grep -v "^+" | cut -c1-250; cat $OUT/celeba_ddim250.json; tail -5 $OUT/gputests.txt
