# round 3, session D: library-side Fri.commit + non-blocking frees: GPU suite, leak / stress tools, Fri.prove timing
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
(timeout 1700 python -m pytest tests -x -q -m gpu 2>&1 | tail -15) > $O/pytest_gpu.txt
(timeout 600 python tools/leak_check.py > $O/leak_check.txt) 2>&1
(timeout 600 python tools/fri_stress.py > $O/fri_stress.txt) 2>&1
(timeout 600 python tools/sharded_fri_profile.py 22 8 2>&1 | head -40) > $O/sharded_fri_profile.txt
(timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 2>$O/bench_s20.err | tail -1) > $O/bench_s20.json
cat $O/pytest_gpu.txt; cat $O/leak_check.txt; tail -3 $O/fri_stress.txt; head -4 $O/sharded_fri_profile.txt
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench_s20.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['frac'], d['clock_ramp']['steady_state']['value'])
e=d['extras']
print({k:(v.get('ms') or v.get('ms_per_pair')) for k,v in e.items() if isinstance(v,dict)})
print(e['fri_prove_2p22_ef4_s40'])
print(e.get('ntt_2p24_strong'))
PY
