#!/bin/bash
# GPU idle gaps of one FastStark.prove at FRI 2^24 out of a loop of proofs (rocprofv3 --kernel-trace -> tools/gap_report.py).
#   gpurun -- 'bash tools/gpu_stark_gaps.sh <outdir> [eager]'
O=gpurun_out/$1; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/tr -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/plain_stark_loop.py 24 8 $2 > $GRAFT_REPO_ROOT/$O/run_$2.txt 2>&1)
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
OFFSET_MS=${OFFSET_MS:-60} WINDOW_MS=${WINDOW_MS:-17.5} python tools/gap_report.py $f 0 15 > $O/stark_prove_plain_2p24_gpu_gaps_$2.txt; rm -rf $O/tr
tail -1 $O/run_$2.txt; head -24 $O/stark_prove_plain_2p24_gpu_gaps_$2.txt
