O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_host.py tests/test_gpu_cabi.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -5) > $O/pytest.txt
(timeout 300 python tools/fri_phase2.py 2>&1 | head -3) > $O/fri_phase2.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace -o fri --output-format csv -- python $GRAFT_REPO_ROOT/tools/fri_trace_run.py > $GRAFT_REPO_ROOT/$O/fri_trace_run.txt 2>&1)
f=$(ls $O/trace/*kernel_trace.csv | head -1); python tools/fri_trace_report.py $f > $O/fri_timeline.txt
rm -rf $O/trace
(timeout 300 python tools/fri_stress.py 10 > $O/fri_stress.txt) 2>&1
cat $O/pytest.txt; cat $O/fri_phase2.txt | head -2; tail -40 $O/fri_timeline.txt; tail -2 $O/fri_stress.txt
