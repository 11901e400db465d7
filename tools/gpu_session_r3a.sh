# round 3, session A: baseline timeline of the sharded path at world 1 over RCCL + headline in the driver-style window
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python tools/sharded_timeline.py run 21 200 > $O/sharded_timeline_before.json) 2> $O/sharded_timeline_before.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o st --output-format csv -- python $GRAFT_REPO_ROOT/tools/sharded_timeline.py run 21 60 > $GRAFT_REPO_ROOT/$O/trace_run.json 2> $GRAFT_REPO_ROOT/$O/trace_run.err)
f=$(ls $O/trace/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$f" ] && python tools/sharded_timeline.py report $f > $O/sharded_timeline_before.txt
# keep the merged output small: the raw trace can be tens of MB
ls -la $O/trace > $O/trace_ls.txt 2>&1; find $O/trace -name '*kernel_trace.csv' -size +20M -delete
(timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>$O/bench_s20.err | tail -1) > $O/bench_s20.json
cat $O/sharded_timeline_before.json; head -40 $O/sharded_timeline_before.txt; cat $O/bench_s20.json | cut -c1-400
