O=gpurun_out/r2i; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $O/gpu_tests.txt
(timeout 300 python tools/ab3.py '{"prio_balance":0}' 2>&1 | grep -v amdgpu.ids) > $O/ab3.txt
tail -3 $O/gpu_tests.txt; cat $O/ab3.txt
