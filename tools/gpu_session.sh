mkdir -p gpurun_out/r2c
(timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -15) > gpurun_out/r2c/gpu_tests.txt
(timeout 300 python tools/ab3.py '{"looped":0}' '{"prune":0}' '{"looped":0,"prune":0}' '{"wave_local":0}' 2>&1 | grep -v amdgpu.ids) > gpurun_out/r2c/ab3_main.txt
(STARKCORE_LIB=$PWD/stark-anatomy_amd/libstarkcore_prio.so timeout 200 python tools/ab3.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r2c/ab3_prio.txt
(STARKCORE_LIB=$PWD/stark-anatomy_amd/libstarkcore_w8.so timeout 200 python tools/ab3.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/r2c/ab3_w8.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $GRAFT_REPO_ROOT/gpurun_out/r2c/counters.txt 2>&1
cd $GRAFT_REPO_ROOT
for ctr in "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM"; do
  tag=$(echo $ctr | tr ' ' '_' | cut -c1-40)
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $GRAFT_REPO_ROOT/gpurun_out/r2c/pmc_$tag -o pmc --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --log2n 22 > $GRAFT_REPO_ROOT/gpurun_out/r2c/pmc_$tag.log 2>&1)
done
tail -3 gpurun_out/r2c/gpu_tests.txt; cat gpurun_out/r2c/ab3_main.txt
