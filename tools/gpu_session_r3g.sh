O=gpurun_out/r3g; mkdir -p $O
export TMPDIR=/tmp
(timeout 300 python tools/ab3.py > $O/ab_base.txt) 2>&1
(STARKCORE_LIB=$PWD/stark-anatomy_amd/libstarkcore_nt.so timeout 300 python tools/ab3.py > $O/ab_twd_nt.txt) 2>&1
(timeout 300 python tools/sharded_stage_timing.py > $O/sharded_stage_timing.txt) 2>&1
(timeout 900 python bench.py --gpus 1 --workload stark_prove --steps 3 --warmup 1 2>$O/stark_prove.err | tail -1) > $O/stark_prove_2p20.json
(timeout 900 python bench.py --gpus 1 --workload stark_prove --log2n 16 --steps 3 --warmup 1 2>>$O/stark_prove.err | tail -1) > $O/stark_prove_2p16.json
cat $O/ab_base.txt $O/ab_twd_nt.txt; cat $O/sharded_stage_timing.txt; cut -c1-900 $O/stark_prove_2p20.json; echo; cut -c1-600 $O/stark_prove_2p16.json; tail -3 $O/stark_prove.err
