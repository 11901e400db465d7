O=gpurun_out/lde; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_cabi.py -x -q -m gpu -k "ntt or coset" 2>&1 | tail -2
for a in "18 3 2" "18 3 4" "18 3 8" "18 3 16" "16 3 16" "20 2 4" "20 4 2" "14 3 64"; do timeout 300 python tools/lde_columns.py $a 2>&1 | grep LDE | tee -a $O/lde.txt; done
for which in old new old new; do
  if [ $which = old ]; then export STARKCORE_LIB=$GRAFT_REPO_ROOT/stark-anatomy_amd/libstarkcore_ab.so; else unset STARKCORE_LIB; fi
  echo "== $which"; for a in "20 64 30" "24 1 20" "20 1 200"; do timeout 300 python tools/columns_ntt.py $a 2>&1 | grep "columns call, one stream"; done; done | tee $O/ab.txt
