O=gpurun_out/e8c; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_cabi.py -x -q -m gpu -k "ntt" 2>&1 | tail -2
for i in 1 2; do for t in 2 3; do
 timeout 600 python bench.py --no-extras --no-cpu-baseline --tune loge_cols=$t 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('loge_cols', $t, round(d['value']/1e9,2), round(d['roofline']['avg_launch_us'],1), 'steady', round(d['clock_ramp']['steady_state']['value']/1e9,2), d['config']['roundtrip_bit_exact'], d['config']['forward_sha256_equals_reference_output'])" | tee -a $O/ab.txt
done; done
for t in "loge_cols=2" "loge_cols=3"; do for a in "17 64 30" "18 64 30" "19 32 30" "20 16 30" "20 8 30"; do echo "TUNE=$t" >> $O/columns.txt; TUNE=$t timeout 300 python tools/columns_ntt.py $a 2>&1 | grep "columns call, one stream\|WRONG" >> $O/columns.txt; done; done; cat $O/columns.txt
