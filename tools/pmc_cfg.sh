#!/bin/bash
# HBM traffic per kernel of the generic rollout at cfg3 (PMC FETCH_SIZE / WRITE_SIZE, separate passes); GPU box, repo root.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_cfg3/p$i -- python3 $R/tools/config_bench.py --configs 3 --steps 1 > $R/gpurun_out/pmc_cfg3_$i.log 2>&1 || exit 1
done
