# Round-6 measurement pass (run on the GPU box through gpurun): the bench line, rocprof kernel stats of the train steps (two streams / one stream),
# PMC traffic, and -- new this round -- the SAMPLER: kernel table and PMC traffic of the DDIM-50 CFG call the bench times (CIFAR-10 128 images = 256
# UNet rows per reverse step; CelebA 128 images).  Summaries are copied into profiles/ by profiles/parse_rocprof.py afterwards.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r6}
mkdir -p $OUT
WHAT=${2:-all}
B="--no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline --no-fp32-ab --no-calibration"
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
timeout 1500 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cifar -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $B > $OUT/prof_cifar.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_celeba -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 3 --warmup 1 $B > $OUT/prof_celeba.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 $B > $OUT/pmc_write.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_celeba -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 1 --warmup 1 $B > $OUT/pmc_fetch_celeba.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_celeba -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 1 --warmup 1 $B > $OUT/pmc_write_celeba.log 2>&1
cd $GRAFT_REPO_ROOT
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof1s ]; then
cd /tmp && export TMPDIR=/tmp
export VD_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cifar_1s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 $B > $OUT/prof_cifar_1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_celeba_1s -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 3 --warmup 1 $B > $OUT/prof_celeba_1s.log 2>&1
unset VD_WGRAD_STREAM
cd $GRAFT_REPO_ROOT
fi
if [ "$WHAT" = all ] || [ "$WHAT" = sampler ]; then
cd /tmp && export TMPDIR=/tmp
# the sampler call of the bench line: DDIM-50, w = 1, 128 images = 256 UNet rows per reverse step (one warm pass + one profiled pass: 2 x 50 steps)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sampler -- python3 $GRAFT_REPO_ROOT/tests/probe/sample_only.py 50 128 cifar10 2 > $OUT/prof_sampler.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_sampler -- python3 $GRAFT_REPO_ROOT/tests/probe/sample_only.py 4 128 cifar10 1 > $OUT/pmc_fetch_sampler.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_sampler -- python3 $GRAFT_REPO_ROOT/tests/probe/sample_only.py 4 128 cifar10 1 > $OUT/pmc_write_sampler.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sampler_celeba -- python3 $GRAFT_REPO_ROOT/tests/probe/sample_only.py 10 128 celeba 2 > $OUT/prof_sampler_celeba.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_sampler_celeba -- python3 $GRAFT_REPO_ROOT/tests/probe/sample_only.py 2 128 celeba 1 > $OUT/pmc_fetch_sampler_celeba.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_sampler_celeba -- python3 $GRAFT_REPO_ROOT/tests/probe/sample_only.py 2 128 celeba 1 > $OUT/pmc_write_sampler_celeba.log 2>&1
cd $GRAFT_REPO_ROOT
fi
if [ "$WHAT" = all ] || [ "$WHAT" = ddim250 ]; then
timeout 900 python tests/probe/celeba_ddim250.py > $OUT/celeba_ddim250.json 2> $OUT/celeba_ddim250.err
fi
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls -R $OUT | head -80
du -sh $OUT
