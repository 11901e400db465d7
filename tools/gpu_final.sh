#!/bin/bash
# The last GPU session of a round on the final sources: the whole -m gpu suite, smoke(), the driver's bench command, the default
# bench run, and `bench.py --gpus 8` with eight ranks sharing the one GPU (wall time beside the driver's 1800 s limit).
#   gpurun -- 'bash tools/gpu_final.sh <outdir>'     then copy what should be judged from gpurun_out/<outdir>/ into profiles/
O=gpurun_out/${1:-final}; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests_all_end_of_round.log 2>&1; tail -3 $O/gpu_tests_all_end_of_round.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke_end_of_round.log 2>&1; tail -1 $O/smoke_end_of_round.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default_steps20_driver_style.json 2> $O/bench_s20.err
timeout 900 python bench.py --no-extras --no-cpu-baseline > $O/bench_default_2000steps.json 2> $O/bench_default.err
timeout 900 python bench.py --no-extras --no-cpu-baseline --columns 1 > $O/bench_one_column_2000steps.json 2> $O/bench_default_1col.err
s=$(date +%s); timeout 1750 python bench.py --gpus 8 > $O/bench_8ranks_sharing_one_gpu_functional.json 2> $O/bench_8ranks.err
echo "bench.py --gpus 8, eight ranks sharing the one GPU (gloo): rc $? wall_s $(( $(date +%s)-s )) (driver limit 1800 s)" | tee $O/bench_8ranks_sharing_one_gpu_wall.txt
python - "$O" <<'PY'
import json, sys
O = sys.argv[1]
for f in ("bench_default_steps20_driver_style.json", "bench_default_2000steps.json", "bench_one_column_2000steps.json", "bench_8ranks_sharing_one_gpu_functional.json"):
    try:
        d = json.loads(open(O + "/" + f).read().strip().splitlines()[-1])
        print(f, round(d["value"] / 1e9, 3), "G el/s", d["ms_per_step"], "ms/step frac", round(d["roofline"]["frac"], 4), d.get("config", {}).get("legs", ""))
    except Exception as e:
        print(f, "ERR", e)
PY
