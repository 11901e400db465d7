#!/bin/bash
# One GPU-box session, parameterised: `gpurun -- 'bash tools/gpu_steps.sh <outdir-name> <step> [<step> ...]'`.
# Every step writes under gpurun_out/<outdir-name>/ (merged back by gpurun); copy what should be judged into profiles/.
#   tests            the whole -m gpu suite                      tests:<file>      one test file
#   bench            python bench.py (default window)            bench:<args>      bench.py with arguments (comma = space)
#   rocprof:<args>   rocprofv3 --kernel-trace --stats of bench.py <args>; keeps the kernel_stats.csv head and the JSON line
#   prover:<log>     tools/sharded_stark_profile.py <log>        timeline:<forms>  tools/sharded_timeline.py under --kernel-trace
#   py:<script>      python tools/<script>.py           ubench:<name>     build and run tools/microbench/<name>.hip
O=gpurun_out/$1; shift; mkdir -p $O
export TMPDIR=/tmp
for step in "$@"; do
  name=${step%%:*}; arg=${step#*:}; [ "$arg" = "$step" ] && arg=""
  arg=${arg//,/ }
  tag=$(echo "$step" | tr -c 'A-Za-z0-9_.-' '_')
  case $name in
    tests)   target=tests; [ -n "$arg" ] && target=tests/$arg
             (timeout 3000 python -m pytest $target -x -q -m gpu 2>&1 | tail -25) > $O/pytest_$tag.txt; cat $O/pytest_$tag.txt ;;
    bench)   (timeout 1500 python bench.py $arg > $O/bench_$tag.json 2> $O/bench_$tag.err); cut -c1-400 $O/bench_$tag.json ;;
    rocprof) (cd /tmp && timeout 1500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_$tag -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py $arg > $GRAFT_REPO_ROOT/$O/rocprof_$tag.txt 2>&1)
             f=$(ls $O/trace_$tag/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -45 $f > $O/kernel_stats_$tag.csv
             rm -rf $O/trace_$tag; grep '^{' $O/rocprof_$tag.txt > $O/bench_under_rocprof_$tag.json; head -8 $O/kernel_stats_$tag.csv | cut -c1-170 ;;
    prover)  (timeout 1500 python tools/sharded_stark_profile.py $arg > $O/prover_$tag.txt 2>&1); head -24 $O/prover_$tag.txt ;;
    timeline) (cd /tmp && TIMELINE_FORMS=${arg// /,} timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/tl_$tag -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/sharded_timeline.py run 21 200 > $GRAFT_REPO_ROOT/$O/timeline_run_$tag.txt 2>&1)
             f=$(ls $O/tl_$tag/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/sharded_timeline.py report $f > $O/timeline_$tag.txt; rm -rf $O/tl_$tag
             grep '^{' $O/timeline_run_$tag.txt; head -30 $O/timeline_$tag.txt ;;
    ubench)  (cd tools/microbench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -Wno-unused-result -o /tmp/ub_$arg $arg.hip 2>&1 | grep -E "error" ; timeout 300 /tmp/ub_$arg) > $O/ubench_$tag.txt 2>&1; cat $O/ubench_$tag.txt ;;
    py)      s=${arg%% *}; rest=${arg#"$s"}; (timeout 1500 python tools/$s.py $rest > $O/py_$tag.txt 2>&1); tail -40 $O/py_$tag.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
