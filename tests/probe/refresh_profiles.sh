set -x
mkdir -p gpurun_out/r2f
OUT=$GRAFT_REPO_ROOT/gpurun_out/r2f
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $OUT/gputests.txt
timeout 900 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cifar -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-sample --no-cpu-baseline --no-secondary --no-extras > $OUT/prof_cifar.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_celeba -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 3 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras > $OUT/prof_celeba.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras > $OUT/pmc_write.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_trace.csv" -delete
ls -R $OUT | head -40
du -sh $OUT
cat $OUT/gputests.txt
