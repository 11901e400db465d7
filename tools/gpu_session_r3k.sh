O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lg in 20 24; do
  (cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $R/$O/pmc_SQ_INSTS_VALU_SQ_WAVES_S_$lg -o p --output-format csv -- python $R/bench.py --no-extras --no-cpu-baseline --log2n $lg --steps 10 --warmup 2 > /dev/null 2>$R/$O/pmc_$lg.err)
done
python tools/pmc_summary.py $O/pmc_* > $O/pmc_summary.json
find $O -name '*counter_collection.csv' -delete
(timeout 300 python tools/ab3.py > $O/ab_base.txt) 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/r3k/pmc_summary.json'))
for k,v in d.items():
    print(k, {kk[:40]: {c: round(x['avg_per_dispatch']) for c,x in vv.items()} for kk,vv in v.items() if 'ntt_pass' in kk})
PY
cat $O/ab_base.txt
