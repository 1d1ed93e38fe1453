# Round-4 first measurement pass (GPU box): A/B of the F(4x4,3x3) input gradient with plain ds_read_b64 patch reads against the round-3
# build (ds_read2_b64), its LDS counters, the new parity tests, and a baseline bench line of this box.
set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-r4a}
mkdir -p $OUT
L=$GRAFT_REPO_ROOT/v-diffusion-torch_amd/lib
for i in 1 2; do
  echo "== product (ds_read_b64)"; python tests/perf_wino43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/'
  echo "== round 3 (ds_read2_b64)"; VDIFF_HIP_LIB=$L/exp/libw43_r03.so python tests/perf_wino43.py 2>&1 | grep "F(4,3)" | sed 's/.*| F(4,3)/F(4,3)/'
done > $OUT/w43_ab.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_wino2 -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/pmc_wino2.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OUT/pmc_wino -- python3 $GRAFT_REPO_ROOT/tests/probe/wino_pmc_target.py > $OUT/pmc_wino.log 2>&1
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_bench_shapes_gpu.py tests/test_multigpu_path_gpu.py -q -m gpu -s -x -k "at_bench_launches or 256_rows or b128_rows or two_ranks_equal" 2>&1 | tail -60 > $OUT/newtests.txt
timeout 1200 python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*agent_info.csv" -delete
cat $OUT/w43_ab.txt
tail -30 $OUT/newtests.txt
