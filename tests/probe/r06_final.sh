# Round-6 closing pass (GPU box): item-budget stamps of the convolution kernel, its PMC counters, the GPU suite, the bench line.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6f
mkdir -p $OUT
python tests/probe/w43_phases.py > $OUT/phases.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_wino -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/pmc_wino.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_wino2 -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/pmc_wino2.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 2600 python -m pytest tests -q -m gpu --durations=10 2>&1 | tail -40 > $OUT/gputests.txt
timeout 1500 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python __graft_entry__.py smoke > $OUT/smoke.txt 2>&1
tail -15 $OUT/gputests.txt; tail -2 $OUT/smoke.txt; head -c 300 $OUT/bench_n1.json
