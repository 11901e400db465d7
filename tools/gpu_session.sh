O=gpurun_out/r2m; mkdir -p $O
R=$GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/gpu_tests.txt
(timeout 300 python tools/ab3.py '{"prio_balance":0}' '{"wave_local":0}' '{"prune":0}' 2>&1 | grep -v amdgpu.ids) > $O/ab3.txt
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>&1 | tail -1) > $O/bench_default_steps20.json
cd /tmp && export TMPDIR=/tmp
(timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 100 --warmup 10 --no-extras --no-cpu-baseline > $R/$O/bench_2p20_under_rocprof.json 2> $R/$O/prof_bench.log)
(timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_bench24 -o bench --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --log2n 24 > $R/$O/bench_2p24_under_rocprof.json 2> $R/$O/prof_bench24.log)
for lg in 20 22 24; do for ctr in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-24)_$lg
  (timeout 300 rocprofv3 --pmc $ctr -d $R/$O/pmc_$tag -o pmc --output-format csv -- python $R/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --log2n $lg > /dev/null 2> $R/$O/pmc_$tag.log)
done; done
cd $R
python tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json 2>/dev/null
(timeout 100 python tools/pass_trace.py 20 2>&1 | grep -v amdgpu.ids) > $O/pass_trace_2p20.txt
rm -rf $O/pmc_*/*/ 2>/dev/null
tail -2 $O/gpu_tests.txt; cat $O/ab3.txt; cut -c1-420 $O/bench_default_steps20.json; head -2 $O/prof_bench/bench_kernel_stats.csv | cut -c1-160
