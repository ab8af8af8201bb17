#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6; mkdir -p $O
BA="--no-d256 --no-f32 --no-cpu-baseline --train-steps 0 --no-query-gmm --steps 30 --warmup 5 --sustain-s 1"
for v in old new old new; do
  lib=$PWD/aline_amd/csrc/variants/lib_bufdma.so; [ $v = new ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 200 python bench.py $BA 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', 'headline ms', round(d['ms_per_step'],3), 'sustained', round(d['sustained_ms_per_step'],3), 'kernel', d['roofline']['kernel'], round(d['roofline']['kernel_ms_per_launch']*1e3,1),'us')" | tee -a $O/s3_ab.txt
done
for v in old new; do
  lib=$PWD/aline_amd/csrc/variants/lib_bufdma.so; [ $v = new ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 300 python tools/config_bench.py --configs 1,3 --precs f16x3 2>> $O/cfg.err | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v', d.get('config'), d.get('path'), round(d['ms_per_rollout'],3))" | tee -a $O/s3_ab.txt
done
timeout -k 10 900 python -m pytest tests/test_s3_gpu.py tests/test_hip_parity.py tests/test_fullsize_gpu.py tests/test_edge_cases_gpu.py tests/test_r2_gpu.py -m gpu -q --durations=8 > $O/tests.log 2>&1; echo "s3 tests rc=$?"; tail -16 $O/tests.log
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_cfg3train -- python3 $GRAFT_REPO_ROOT/tools/train_cfg3.py > $GRAFT_REPO_ROOT/$O/train_cfg3.log 2>&1); echo "train cfg3 prof rc=$?"; tail -2 $O/train_cfg3.log
python3 tools/prof_stats.py $O/prof_cfg3train 22
