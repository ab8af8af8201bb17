#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s7; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_backward_gpu.py -m gpu -q -x --durations=6 > $O/tests.log 2>&1; echo "bwd tests rc=$?"; tail -14 $O/tests.log
timeout -k 10 300 python tools/train_cfg3.py > $O/train_cfg3.log 2>&1; echo "train cfg3 rc=$?"; tail -2 $O/train_cfg3.log
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg3train -- python3 $GRAFT_REPO_ROOT/tools/train_cfg3.py > $GRAFT_REPO_ROOT/$O/train_cfg3_prof.log 2>&1); echo "train cfg3 prof rc=$?"
python3 tools/prof_stats.py $O/prof_cfg3train 14
