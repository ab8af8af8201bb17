# timing experiments on the generic f16x3 GEMM (GPU box): cfg5 rollout time with parts of gemm_nt_kernel<3,...> removed
# (results are garbage, only the time is read).  Build first: tools/x3_variants.sh build "gnomfma:-DGEMM_NO_MFMA" ...
for v in base gnomfma gnogload gnosplit gnone; do
  lib=aline_amd/csrc/variants/lib_$v.so; [ $v = base ] && lib=aline_amd/csrc/libaline_hip.so
  echo "$v: $(ALINE_HIP_LIB=$lib python tools/config_bench.py --configs 5 --precs f16x3 2>/dev/null | head -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_rollout"],2))')"
done
