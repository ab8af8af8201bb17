#!/bin/bash
# session 12: per-kernel times of the x5 path at cfg5
cd $GRAFT_REPO_ROOT
O=gpurun_out/s12; mkdir -p $O
for F in 128 2048; do
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg5_$F -- python3 $GRAFT_REPO_ROOT/tools/cfg5_run.py f16x3 $F > $GRAFT_REPO_ROOT/$O/cfg5_$F.log 2>&1); echo "cfg5 F=$F prof rc=$?"
python3 tools/prof_stats.py $O/prof_cfg5_$F 14
done
