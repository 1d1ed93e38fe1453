set -x
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4s
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_sampler -- python3 $GRAFT_REPO_ROOT/tests/probe/sample_only.py > $OUT/prof_sampler.log 2>&1
cd $GRAFT_REPO_ROOT
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls $OUT/prof_sampler/*/
