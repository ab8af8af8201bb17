#!/bin/bash
# HBM traffic of the training-step kernels (PMC FETCH_SIZE / WRITE_SIZE, separate passes); GPU box, repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train/p$i -- python3 $R/bench.py --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --train-steps 1 > $R/gpurun_out/pmc_train_$i.log 2>&1 || exit 1
done
