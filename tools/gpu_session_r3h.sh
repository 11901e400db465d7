O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
(timeout 900 python bench.py --gpus 1 --workload stark_prove --log2n 16 --steps 3 --warmup 1 2>$O/stark_prove.err | tail -1) > $O/stark_prove_2p16.json
(timeout 900 python bench.py --gpus 1 --workload stark_prove --steps 2 --warmup 1 2>>$O/stark_prove.err | tail -1) > $O/stark_prove_2p20.json
cut -c1-1200 $O/stark_prove_2p16.json; echo; cut -c1-1200 $O/stark_prove_2p20.json; grep -v "amdgpu.ids\|socket.cpp" $O/stark_prove.err | tail
