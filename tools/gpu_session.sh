O=gpurun_out/r2h; mkdir -p $O
(timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $O/gpu_tests.txt
(timeout 300 python tools/ab3.py 2>&1 | grep -v amdgpu.ids) > $O/ab3_mul2.txt
(STARKCORE_LIB=$PWD/stark-anatomy_amd/libstarkcore_nomul2.so timeout 300 python tools/ab3.py 2>&1 | grep -v amdgpu.ids) > $O/ab3_nomul2.txt
(timeout 300 python tools/ab3.py 2>&1 | grep -v amdgpu.ids) > $O/ab3_mul2_again.txt
tail -3 $O/gpu_tests.txt; cat $O/ab3_mul2.txt $O/ab3_nomul2.txt $O/ab3_mul2_again.txt
