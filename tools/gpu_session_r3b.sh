# round 3, session B: the plan-object sharded path: GPU tests, timeline at world 1 (in place / through RCCL, torch / native)
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_sharded.py -x -q -m gpu 2>&1 | tail -25) > $O/pytest_sharded.txt
(timeout 300 python tools/sharded_timeline.py run 21 200 > $O/sharded_timeline.json) 2> $O/sharded_timeline.err
(cd /tmp && TIMELINE_FORMS=in_place_nothing_to_exchange,native_own_block_through_rccl,native_own_block_through_rccl_4_blocks timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o st --output-format csv -- python $GRAFT_REPO_ROOT/tools/sharded_timeline.py run 21 40 > $GRAFT_REPO_ROOT/$O/trace_run.json 2> $GRAFT_REPO_ROOT/$O/trace_run.err)
find $O/trace -name '*kernel_trace.csv' -size +30M -delete
cat $O/pytest_sharded.txt; cat $O/sharded_timeline.json; tail -5 $O/sharded_timeline.err
