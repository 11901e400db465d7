O=gpurun_out/r3ab; mkdir -p $O
for i in 1 2; do
(STARKCORE_LIB=$PWD/stark-anatomy_amd/libstarkcore_ab_old.so timeout 300 python tools/ab3.py > $O/ab_old_$i.txt) 2>&1
(timeout 300 python tools/ab3.py > $O/ab_new_$i.txt) 2>&1
done
for f in $O/ab_old_1.txt $O/ab_new_1.txt $O/ab_old_2.txt $O/ab_new_2.txt; do echo $f; cat $f; done
