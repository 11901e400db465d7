O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python tools/fri_phase2.py 2>&1 | head -30) > $O/fri_phase2.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace -o fri --output-format csv -- python $GRAFT_REPO_ROOT/tools/fri_trace_run.py > $GRAFT_REPO_ROOT/$O/fri_trace_run.txt 2>&1)
f=$(ls $O/trace/*kernel_trace.csv | head -1); python tools/fri_trace_report.py $f > $O/fri_timeline.txt
rm -rf $O/trace
cat $O/fri_phase2.txt | head -3; tail -50 $O/fri_timeline.txt; tail -3 $O/fri_trace_run.txt
