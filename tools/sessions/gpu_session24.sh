#!/bin/bash
# session 24: cfg3 training step with / without the saved activations + kernel stats
cd $GRAFT_REPO_ROOT
O=gpurun_out/s24; mkdir -p $O
timeout -k 10 300 python tools/train_cfg3.py 2>&1 | tail -3
ALINE_DBG=NO_BWD_SAVED_ACTS timeout -k 10 300 python tools/train_cfg3.py 2>&1 | tail -3
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/tools/train_cfg3.py > $GRAFT_REPO_ROOT/$O/run.log 2>&1); echo "rc=$?"
python3 tools/prof_stats.py $O/prof 16
