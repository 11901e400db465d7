O=gpurun_out/r2k; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $O/gpu_tests.txt
(timeout 300 python tools/ab3.py 2>&1 | grep -v amdgpu.ids) > $O/ab3.txt
(timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400) > $O/bench20.txt
tail -3 $O/gpu_tests.txt; cat $O/ab3.txt $O/bench20.txt
