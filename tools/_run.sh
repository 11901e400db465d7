O=gpurun_out/dtw; mkdir -p $O
for t in "" "direct_tw_max_log=24" "" "direct_tw_max_log=24"; do
for a in "24 1 20" "23 1 20"; do echo "TUNE=$t" >> $O/columns.txt; TUNE=$t timeout 300 python tools/columns_ntt.py $a 2>&1 | grep "one at a time, one stream\|WRONG\|equal" >> $O/columns.txt; done; done; cat $O/columns.txt
