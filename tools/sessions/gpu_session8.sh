#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s8; mkdir -p $O
BA="--no-d256 --no-f32 --no-cpu-baseline --train-steps 0 --no-query-gmm --steps 30 --warmup 5 --sustain-s 1"
for v in prev new prev new; do
  lib=$PWD/aline_amd/csrc/variants/lib_prev.so; [ $v = new ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 200 python bench.py $BA 2>> $O/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$v', 'headline ms', round(d['ms_per_step'],3), 'sustained', round(d['sustained_ms_per_step'],3), 'kernel', round(d['roofline']['kernel_ms_per_launch']*1e3,1),'us')" | tee -a $O/s3_ab.txt
done
for v in prev new; do
  lib=$PWD/aline_amd/csrc/variants/lib_prev.so; [ $v = new ] && lib=$PWD/aline_amd/csrc/libaline_hip.so
  ALINE_HIP_LIB=$lib timeout -k 10 300 python tools/config_bench.py --configs 3 --precs f16x3 2>> $O/cfg.err | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('$v', d.get('config'), d.get('path'), round(d['ms_per_rollout'],3))" | tee -a $O/s3_ab.txt
  ALINE_HIP_LIB=$lib timeout -k 10 120 python tools/x3_time.py 2>> $O/x3_time.err | tee -a $O/s3_ab.txt
done
timeout -k 10 600 python -m pytest tests/test_s3_gpu.py tests/test_x3_gpu.py tests/test_hip_parity.py tests/test_r2_gpu.py tests/test_fullsize_gpu.py -m gpu -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/tests.log
