# Round-5 measurement pass (run on the GPU box through gpurun): tests, the bench line, rocprof kernel stats, PMC traffic, the in-kernel
# clock stamps of the Winograd kernels and one PMC pass over them.  Summaries are copied into profiles/ by hand afterwards.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r5}
mkdir -p $OUT
WHAT=${2:-all}
if [ "$WHAT" = all ] || [ "$WHAT" = tests ]; then
timeout 2400 python -m pytest tests -q -m gpu -s 2>&1 | tail -80 > $OUT/gputests.txt
fi
if [ "$WHAT" = all ] || [ "$WHAT" = bench ]; then
timeout 1200 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof ]; then
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cifar -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/prof_cifar.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_celeba -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 3 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/prof_celeba.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/pmc_write.log 2>&1
# CelebA (BASELINE configs[3]) traffic passes: the secondary block's `traffic` must come from a CelebA run or be null
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_celeba -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 1 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/pmc_fetch_celeba.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_celeba -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 1 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/pmc_write_celeba.log 2>&1
# the two dominant Winograd kernels: MFMA-busy / VALU / wait / active cycles (SQ: 8 slots, GRBM: 2), one pass
rocprofv3 -L > $OUT/counters_available.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_wino -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/pmc_wino.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_wino2 -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/pmc_wino2.log 2>&1
cd $GRAFT_REPO_ROOT
fi
if [ "$WHAT" = all ] || [ "$WHAT" = prof1s ]; then
# the same two workloads with the weight gradients on the main stream (VD_WGRAD_STREAM=0): every kernel runs alone, so the average
# durations are the kernels' own (with the side stream a launch's duration includes the time it shares the chip with the other stream)
cd /tmp && export TMPDIR=/tmp
export VD_WGRAD_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_cifar_1s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/prof_cifar_1s.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_celeba_1s -- python3 $GRAFT_REPO_ROOT/bench.py --config celeba --steps 3 --warmup 1 --no-sample --no-cpu-baseline --no-secondary --no-extras --no-torch-baseline > $OUT/prof_celeba_1s.log 2>&1
unset VD_WGRAD_STREAM
cd $GRAFT_REPO_ROOT
fi
if [ "$WHAT" = all ] || [ "$WHAT" = clock ]; then
# in-kernel clock stamps (probe library): cycles per K tile and MHz of the Winograd forward / weight-gradient kernels
VD_WINO_PROBE_LIGHT=1 timeout 600 python tests/probe/wino_phases.py > $OUT/wino_phases_light.txt 2>&1
timeout 600 python tests/probe/wgrad_clock.py > $OUT/wgrad_clock.txt 2>&1
# the clock the chip HOLDS beside each kernel class and beside the whole step: a witness kernel on a second stream (product library, no stamps in it)
timeout 900 python tests/probe/clock_by_kernel.py > $OUT/clock_by_kernel.txt 2> $OUT/clock_by_kernel.err
fi
if [ "$WHAT" = all ] || [ "$WHAT" = ddim250 ]; then
timeout 900 python tests/probe/celeba_ddim250.py > $OUT/celeba_ddim250.json 2> $OUT/celeba_ddim250.err
fi
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
ls -R $OUT | head -60
du -sh $OUT
tail -5 $OUT/gputests.txt
