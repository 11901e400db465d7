O=gpurun_out/r2o; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -5) > $O/gpu_tests.txt
(timeout 300 python tools/sharded_stage_timing.py 2>&1 | grep -v amdgpu.ids | head -5) > $O/sharded_stage_timing.txt
tail -2 $O/gpu_tests.txt; cut -c1-520 $O/sharded_stage_timing.txt
