# timing sweep of the s3 launch shape (GPU box): waves per workgroup x episodes per workgroup on cfg2, default on cfg3
run() { python tools/config_bench.py --configs $1 --precs f16x3 2>/dev/null | grep '"d": 32' | head -1 | python -c 'import json,sys; d=json.loads(sys.stdin.readline()); print(round(d["ms_per_rollout"],3), d["path"])'; }
for spec in ${SWEEP:-16:0 16:4 16:5 8:0 8:4}; do
  w=${spec%%:*}; e=${spec#*:}
  echo "cfg2 waves=$w epw=$e: $(ALINE_DBG=S3_WAVES=$w,S3_EPW=$e run 2)"
done
echo "cfg3: $(run 3)"
