#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/full; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q --durations=10 > $O/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -18 $O/gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
