#!/bin/bash
# A/B of an environment switch of the library in ONE session on one box, alternating:  bash tools/ab_env.sh <outdir> <rounds> VAR=off_value <tools script> <args...>
O=gpurun_out/$1; R=$2; KV=$3; shift 3; mkdir -p $O
for i in $(seq 1 $R); do
  for which in off on; do
    if [ $which = off ]; then export "$KV"; else unset "${KV%%=*}"; fi
    echo "== $KV $which (round $i)"; timeout 600 python tools/"$@" 2>&1 | grep -v amdgpu.ids | tail -${TAIL:-3}
  done
done | tee $O/ab_env_$(echo "$KV $*" | tr -c 'A-Za-z0-9_.-' '_').txt
