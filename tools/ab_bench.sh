#!/bin/bash
# same-box A/B of two library builds through bench.py's own legs (Fri.prove 2^22, FastStark.prove 2^24, Merkle 2^24)
O=gpurun_out/ab_bench; mkdir -p $O
for i in 1 2 3; do
  for which in old new; do
    if [ $which = old ]; then export STARKCORE_LIB=$GRAFT_REPO_ROOT/stark-anatomy_amd/libstarkcore_ab.so; else unset STARKCORE_LIB; fi
    timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/${which}_$i.json 2> $O/${which}_$i.err
    python - $O/${which}_$i.json $which $i <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e = d["extras"]
f, s = e["fri_prove_2p22_ef4_s40"], e["stark_prove_2p24_1gpu"]
print("%s (round %s): Fri.prove 2^22 %.3f best / %.3f median ms   FastStark.prove 2^24 %.2f best / %.2f median ms   Merkle 2^24 %.3f ms   headline %.2f G el/s"
      % (sys.argv[2], sys.argv[3], f["ms"], f["median_ms"], s["ms_per_proof"], s["median_ms"], e["merkle_commit_2p24"]["ms"], d["value"] / 1e9))
PY
  done
done | tee $O/ab_bench.txt
