#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s21; mkdir -p $O
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/d256_train_run.py > $GRAFT_REPO_ROOT/$O/run.log 2>&1); echo "rc=$?"
python3 tools/prof_stats.py $O/prof 22
