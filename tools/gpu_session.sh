O=gpurun_out/r2n; mkdir -p $O
(timeout 400 python tools/ab3.py '{"max_tile_log":10}' '{"max_tile_log":10,"max_col_log":2}' '{"max_tile_log":12}' '{"max_tile_log":12,"max_digit_log":9}' 2>&1 | grep -v amdgpu.ids) > $O/ab3_tiles.txt
cat $O/ab3_tiles.txt
