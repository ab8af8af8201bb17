#!/bin/bash
# Backward-pass check on the GPU box: gradient parity tests, then a profiled training probe (t_chunk 30) and its top kernels.
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests/test_backward_gpu.py -x -q > gpurun_out/tb_test.log 2>&1; tail -2 gpurun_out/tb_test.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_tb
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_tb -- python3 $R/tools/train_probe.py 30 > $R/gpurun_out/tb_prof.log 2>&1
grep t_chunk $R/gpurun_out/tb_prof.log
python3 $R/tools/prof_stats.py $R/gpurun_out/prof_tb ${1:-14}
