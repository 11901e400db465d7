R=$GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests_all.log 2>&1; tail -3 $O/gpu_tests_all.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_s20.err | tail -1) > $O/bench_default_steps20_driver_style.json
timeout 300 python tools/stark_phase_compare.py 24 8 > $O/stark_prove_fri2p24_plain_vs_sharded_world1_phases.txt 2>&1
timeout 300 python tools/stark_phase_compare.py 20 8 > $O/stark_prove_fri2p20_plain_vs_sharded_world1_phases.txt 2>&1
bash tools/gpu_fri_timeline.sh $O/ft > /dev/null 2>&1; cp $O/ft/fri_prove_2p22_timeline.txt $O/; rm -rf $O/ft
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/ks -o k --output-format csv -- python $R/tools/plain_stark_loop.py 24 8 > $R/$O/plain_loop.txt 2>&1); f=$(ls $O/ks/*kernel_stats.csv | head -1); cp $f $O/stark_prove_plain_fri2p24_kernel_stats.csv; f=$(ls $O/ks/*kernel_trace.csv | head -1); python tools/gap_report.py $f 16.6 15 > $O/stark_prove_plain_2p24_gpu_gaps.txt; rm -rf $O/ks
STARKCORE_FRI_TIMING=1 timeout 100 python tools/fri_prove_timing.py 22 6 2>&1 | tail -14 > $O/fri_prove_2p22_one_call_phases.txt
grep -h "best" $O/stark_prove_fri2p2*_phases.txt; tail -2 $O/fri_prove_2p22_timeline.txt; head -3 $O/stark_prove_plain_2p24_gpu_gaps.txt; grep prove $O/plain_loop.txt | tail -1
