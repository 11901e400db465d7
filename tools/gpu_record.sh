#!/bin/bash
# The numbers of record of a round: bench lines, rocprofv3 kernel stats of the same commands, PMC passes (separate runs, as the
# microarchitecture guide prescribes).   gpurun -- 'bash tools/gpu_record.sh r04x'   -> gpurun_out/r04x/ ; copy into profiles/r04/
O=gpurun_out/${1:-record}; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
(timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_s20.err | tail -1) > $O/bench_default_steps20_driver_style.json
(timeout 900 python bench.py --no-extras --no-cpu-baseline 2>$O/bench_default.err | tail -1) > $O/bench_default_2000steps.json
(timeout 900 python bench.py --no-extras --no-cpu-baseline --columns 1 2>$O/bench_default_1col.err | tail -1) > $O/bench_one_column_2000steps.json
# kernel stats of the same commands: the headline (2^20 x 64 columns per step), one 2^20 column at a time, one 2^24 column at a time
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt20x64 -o b --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline 2>$R/$O/kt20x64.err | tail -1 > $R/$O/bench_2p20x64_under_rocprof.json)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt20 -o b --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline --columns 1 2>$R/$O/kt20.err | tail -1 > $R/$O/bench_2p20_under_rocprof.json)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/kt24 -o b --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline --log2n 24 --columns 1 --steps 10 --warmup 2 2>$R/$O/kt24.err | tail -1 > $R/$O/bench_2p24_under_rocprof.json)
for lg in 20x64 20 24; do
  f=$(ls $O/kt$lg/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_2p${lg}_kernel_stats.csv
  f=$(ls $O/kt$lg/*domain_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/bench_2p${lg}_domain_stats.csv
  rm -rf $O/kt$lg
done
# PMC passes, one counter set per run (MI355X_MICROARCH.md): keys 20x64 (the headline's launches), 20 and 24 (one column)
for lg in 20x64 20 24; do
  case $lg in 20x64) a="";; 20) a="--columns 1";; 24) a="--log2n 24 --columns 1";; esac
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES"; do
    tag=$(echo $set | tr ' ' '_' | cut -c1-24)
    (cd /tmp && timeout 600 rocprofv3 --pmc $set -d $R/$O/pmc_${tag}_$lg -o p --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline $a --steps 10 --warmup 2 > /dev/null 2>$R/$O/pmc_${tag}_$lg.err)
  done
done
python tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json
rm -rf $O/pmc_*/ 
python - "$O" <<'PY'
import json, sys, glob
O = sys.argv[1]
d = json.load(open(O + "/pmc_summary.json"))
out = {}
for lg in ("20x64", "20", "24"):
    f = w = None
    for run, kernels in d.items():
        if not run.endswith("_" + lg): continue
        for name, c in kernels.items():
            if "ntt_pass" not in name: continue
            if "FETCH_SIZE" in c: f = c["FETCH_SIZE"]["avg_per_dispatch"]
            if "WRITE_SIZE" in c: w = c["WRITE_SIZE"]["avg_per_dispatch"]
    if f is not None and w is not None:
        out[lg] = {"hbm_bytes_per_launch": (2 * f + w) * 1024, "fetch_size_kb": f, "write_size_kb": w,
                   "formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE doubled: gfx950 correction of MI355X_MICROARCH.md)"}
sys.path.insert(0, "tools")
import pmc_records
out["kernel_source_sha256_16"] = pmc_records.kernel_source_digest()      # bench.py flags PMC figures that predate a change to the kernel's sources
json.dump(out, open(O + "/traffic.json", "w"), indent=1)
for f in sorted(glob.glob(O + "/bench*.json")):
    try:
        b = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], round(b["value"] / 1e9, 3), b["ms_per_step"], round(b["roofline"]["frac"], 4), b["roofline"]["avg_launch_us"])
    except Exception as e:
        print(f, "ERR", e)
print(json.dumps(out))
PY
for lg in 20x64 20 24; do head -4 $O/bench_2p${lg}_kernel_stats.csv | cut -c1-160; done
