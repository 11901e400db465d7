#!/bin/bash
# rocprofv3 kernel trace of Fri.prove at 2^22 -> timeline of the last proof.  gpurun -- 'bash tools/gpu_fri_timeline.sh <outdir>'
O=${1:-gpurun_out/fri_timeline}; mkdir -p $O; export TMPDIR=/tmp
(cd /tmp && timeout 900 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/tr -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/fri_profile.py trace-run > $GRAFT_REPO_ROOT/$O/run.txt 2>&1)
f=$(find $O/tr -name "*kernel_trace.csv" | head -1); python tools/fri_profile.py trace-report $f > $O/fri_prove_2p22_timeline.txt; rm -rf $O/tr
grep prove_ms $O/run.txt | tr '\n' ' '; echo; cat $O/fri_prove_2p22_timeline.txt
