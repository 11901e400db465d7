O=gpurun_out/r2f; mkdir -p $O
(timeout 300 python tools/fri_phase2.py 2>&1 | grep -v amdgpu.ids | head -40) > $O/fri_phase2.txt
(timeout 600 python -m pytest tests/test_gpu_sharded.py -m gpu -q -x 2>&1 | tail -5) > $O/gpu_sharded_tests.txt
cat $O/fri_phase2.txt; cat $O/gpu_sharded_tests.txt
