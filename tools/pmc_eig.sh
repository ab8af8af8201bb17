#!/bin/bash
# HBM traffic of the EIG kernels (PMC FETCH_SIZE / WRITE_SIZE in separate passes, kernel-trace only); run on the GPU box
# from the repo root: tools/pmc_eig.sh ; then python tools/pmc_summary.py gpurun_out/pmc_eig <kernel name>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_eig/p$i -- python3 $R/tools/eig_bench.py > $R/gpurun_out/pmc_eig_$i.log 2>&1 || exit 1
done
