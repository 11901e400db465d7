mkdir -p gpurun_out/r2d
(timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -25) > gpurun_out/r2d/gpu_tests.txt
(timeout 300 python tools/ab3.py '{"prio_balance":0}' '{"prio_balance":1}' '{"prune":0}' 2>&1 | grep -v amdgpu.ids) > gpurun_out/r2d/ab3.txt
(timeout 600 python tools/stark_big_trace.py 12 --host-too 2>&1 | grep -v amdgpu.ids | tail -5) > gpurun_out/r2d/stark_big_12.txt
(timeout 600 python tools/stark_big_trace.py 16 2>&1 | grep -v amdgpu.ids | tail -3) > gpurun_out/r2d/stark_big_16.txt
(timeout 300 python tools/stark_timing.py 2>&1 | grep -v amdgpu.ids | tail -3) > gpurun_out/r2d/stark_timing.txt
tail -5 gpurun_out/r2d/gpu_tests.txt; cat gpurun_out/r2d/ab3.txt gpurun_out/r2d/stark_big_12.txt gpurun_out/r2d/stark_big_16.txt gpurun_out/r2d/stark_timing.txt
