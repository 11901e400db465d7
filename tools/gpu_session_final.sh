# end of round: the driver's sequence -- GPU tests, smoke, the bench line
O=gpurun_out/r3z; mkdir -p $O
(timeout 2300 python -m pytest tests -x -q -m gpu 2>&1 | tail -4) | tee $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench.err | tail -1) > $O/bench_s20.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3z/bench_s20.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['config'], d['roofline']['valu_insts_per_launch'], d['roofline']['traffic'])
print({k:(v.get('ms') or v.get('ms_per_pair')) for k,v in d['extras'].items() if isinstance(v,dict)})
PY
