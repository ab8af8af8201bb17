#!/bin/bash
# Timing experiments on the x3 layer kernel: builds variants of the library with parts of the kernel removed
# (results are garbage, only the time is read) into aline_amd/csrc/variants/, to be run with ALINE_HIP_LIB=<variant>.
#   python tools/probes/timing_variants.py build NAME ...    (here, cross-compiling: a variant is a set of text edits applied to
#                                                              a scratch copy of the sources -- the shipped headers carry no switches)
#   tools/x3_variants.sh build "NAME:-DFLAG -DFLAG" ...      (macro-selected builds, e.g. the stamped diagnostic instantiations)
#   tools/x3_variants.sh run NAME ...                        (on the GPU box: prints ms per rollout and the layer kernel time)
cd "$(dirname "$0")/.." || exit 1
mode=$1; shift
mkdir -p aline_amd/csrc/variants
if [ "$mode" = build ]; then
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    (cd aline_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-slp-vectorize -Wno-unused-function $flags -shared -o variants/lib_$name.so aline_hip.hip) &
  done
  wait
else
  cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
  for name in "$@"; do
    lib=$R/aline_amd/csrc/variants/lib_$name.so; [ "$name" = base ] && lib=$R/aline_amd/csrc/libaline_hip.so
    ALINE_HIP_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/var_$name -- python3 $R/bench.py ${X3_ARGS:---d-model 256 --d-ff 1024 --heads 8 --precision f16x3} --steps 2 --warmup 1 --no-cpu-baseline --train-steps 0 > $R/gpurun_out/var_$name.json 2> $R/gpurun_out/var_$name.err || { echo "$name FAILED"; tail -3 $R/gpurun_out/var_$name.err; continue; }
    echo "== $name: $(python3 -c "import json;print(round(json.load(open('$R/gpurun_out/var_$name.json'))['ms_per_step'],2))") ms per rollout"
    python3 $R/tools/prof_stats.py $R/gpurun_out/var_$name 4
  done
fi
