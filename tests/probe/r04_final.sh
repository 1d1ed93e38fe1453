# Round-4 final pass: the whole GPU suite, the default bench line (reads profiles/r04_*traffic.json), the 1-GPU slice of configs[4]
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4z}
mkdir -p $OUT
timeout 1500 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 900 python tests/probe/celeba_ddim250.py > $OUT/celeba_ddim250.json 2> $OUT/celeba_ddim250.err
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -40 > $OUT/gputests.txt
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
tail -3 $OUT/smoke.txt; tail -6 $OUT/gputests.txt; cat $OUT/celeba_ddim250.json
