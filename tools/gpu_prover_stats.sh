#!/bin/bash
# rocprofv3 --kernel-trace --stats of the two provers BASELINE's metric names beside the transforms: Fri.prove at 2^22 (30 proofs) and
# FastStark.prove at FRI 2^24 (8 proofs); keeps the kernel_stats tables.   gpurun -- 'bash tools/gpu_prover_stats.sh <outdir>'
O=gpurun_out/${1:-prover_stats}; mkdir -p $O; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/fri -o t --output-format csv -- python $R/tools/fri_profile.py timing 22 30 > $R/$O/fri_prove_2p22_run.txt 2>&1)
f=$(find $O/fri -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/fri_prove_2p22_kernel_stats.csv; rm -rf $O/fri
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/$O/stark -o t --output-format csv -- python $R/tools/plain_stark_loop.py 24 8 > $R/$O/stark_prove_2p24_run.txt 2>&1)
f=$(find $O/stark -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/stark_prove_2p24_kernel_stats.csv; rm -rf $O/stark
grep -v amdgpu.ids $O/fri_prove_2p22_run.txt | tail -3; head -12 $O/fri_prove_2p22_kernel_stats.csv | cut -c1-150
grep -v amdgpu.ids $O/stark_prove_2p24_run.txt | tail -2; head -14 $O/stark_prove_2p24_kernel_stats.csv | cut -c1-150
