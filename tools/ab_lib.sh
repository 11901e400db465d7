#!/bin/bash
# A/B of two library builds in ONE session on one box: stark-anatomy_amd/libstarkcore_ab.so (the "old" build, made from another
# checkout) against libstarkcore.so, alternating, each through `python tools/<script> <args>`; prints the last lines of every run.
#   gpurun -- 'bash tools/ab_lib.sh <outdir> <rounds> fri_profile.py timing 22 30'
O=gpurun_out/$1; R=$2; shift 2; mkdir -p $O
for i in $(seq 1 $R); do
  for which in old new; do
    if [ $which = old ]; then export STARKCORE_LIB=$GRAFT_REPO_ROOT/stark-anatomy_amd/libstarkcore_ab.so; else unset STARKCORE_LIB; fi
    echo "== $which (round $i)"; timeout 600 python tools/"$@" 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-3}
  done
done | tee $O/ab_$(echo "$*" | tr -c 'A-Za-z0-9_.-' '_').txt
