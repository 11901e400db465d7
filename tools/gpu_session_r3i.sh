O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
(timeout 1500 python -m pytest tests/test_gpu_stark.py tests/test_gpu_sharded.py tests/test_gpu_cabi.py -x -q -m gpu 2>&1 | tail -15) > $O/pytest.txt
(timeout 900 python bench.py --gpus 1 --workload stark_prove --log2n 16 --steps 3 --warmup 1 2>$O/stark_prove.err | tail -1) > $O/stark_prove_2p16.json
(timeout 900 python bench.py --gpus 1 --workload stark_prove --steps 3 --warmup 1 2>>$O/stark_prove.err | tail -1) > $O/stark_prove_2p20.json
(timeout 900 python tools/stark_big_trace.py 16 > $O/stark_big_trace_2p16.txt) 2>&1
cat $O/pytest.txt; python -c "
import json
for f in ('stark_prove_2p16','stark_prove_2p20'):
    d=json.loads(open('gpurun_out/r3i/%s.json'%f).read()); print(f, d['value'], d['config']['runs_ms'], d['config']['verify_accepts'])
"; cat $O/stark_big_trace_2p16.txt | cut -c1-700
