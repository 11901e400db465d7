# round 3: the numbers of record -- bench lines, rocprofv3 kernel stats of the same command, PMC passes (separate runs)
O=gpurun_out/r3j; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_s20.err | tail -1) > $O/bench_default_steps20_driver_style.json
(timeout 900 python bench.py --no-extras --no-cpu-baseline 2>$O/bench_default.err | tail -1) > $O/bench_default_2000steps.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt20 -o b --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline 2>$R/$O/kt20.err | tail -1 > $R/$O/bench_2p20_under_rocprof.json)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt24 -o b --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline --log2n 24 --steps 10 --warmup 2 2>$R/$O/kt24.err | tail -1 > $R/$O/bench_2p24_under_rocprof.json)
find $O -name '*kernel_trace.csv' -delete
for lg in 20 24; do
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-24)
    (cd /tmp && timeout 600 rocprofv3 --pmc $set -d $R/$O/pmc_${tag}_$lg -o p --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline --log2n $lg --steps 10 --warmup 2 > /dev/null 2>$R/$O/pmc_${tag}_$lg.err)
  done
done
python tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json
find $O -name '*counter_collection.csv' -delete
(timeout 600 python bench.py --gpus 1 --force-sharded --steps 200 --warmup 20 --no-extras 2>$O/fs.err | tail -1) > $O/bench_1rank_sharded_path.json
(timeout 600 python bench.py --gpus 1 --force-sharded --force-diag-exchange --steps 200 --warmup 20 --no-extras 2>>$O/fs.err | tail -1) > $O/bench_1rank_sharded_path_own_block_through_rccl.json
(timeout 900 python bench.py --gpus 2 --no-cpu-baseline 2>$O/g2.err | tail -1) > $O/bench_2ranks_sharing_one_gpu_functional.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j/bench*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], round(d['value']/1e9,3), d['ms_per_step'], round(d['roofline']['frac'],4), d['roofline']['avg_launch_us'], d['config'].get('corner_turn','')[:80])
    except Exception as e: print(f, 'ERR', e)
PY
head -c 1500 $O/pmc_summary.json
