#!/bin/bash
# where the GPU idles inside a train step: kernel trace of 8 steps (both streams, then one stream)
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-sample --no-cpu-baseline --no-secondary --no-extras > $OUT/trace.log 2>&1
VD_WGRAD_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace1s -- python3 $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 3 --no-sample --no-cpu-baseline --no-secondary --no-extras > $OUT/trace1s.log 2>&1
cd $GRAFT_REPO_ROOT
python tests/probe/trace_gaps.py $OUT/trace/runc/*_kernel_trace.csv 6 > $OUT/gaps.txt 2>&1
python tests/probe/trace_gaps.py $OUT/trace1s/runc/*_kernel_trace.csv 6 > $OUT/gaps_1s.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/gaps.txt $OUT/gaps_1s.txt
tail -2 $OUT/trace.log | cut -c1-300
