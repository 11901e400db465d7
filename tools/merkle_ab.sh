for rep in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then export STARKCORE_LIB=$GRAFT_REPO_ROOT/stark-anatomy_amd/libstarkcore_ab.so; else unset STARKCORE_LIB; fi
    echo "== $which"; timeout 100 python tools/merkle_timing.py 2>&1 | grep '"logn": 2[024]'
  done
done
