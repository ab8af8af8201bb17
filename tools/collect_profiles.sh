#!/bin/bash
# Round-end evidence run (GPU box, from the repo root): bench line + its rocprofv3 kernel stats, per-config table, s3 PMC
# passes and phase stamps.  Everything lands under gpurun_out/r02/; copy what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || { echo "bench profile failed"; tail -5 $O/bench_prof.err; exit 1; }
echo "bench profile done"
python3 $R/tools/config_bench.py > $O/config_bench.jsonl 2> $O/config_bench.err || { echo "config bench failed"; tail -5 $O/config_bench.err; exit 1; }
echo "config bench done"
cd $R
for c in 2 3; do
  ALINE_HIP_LIB=$R/aline_amd/csrc/variants/lib_s3stamps.so python3 tools/s3_stamps.py $([ $c = 2 ] && echo 30 || echo 50) $c > $O/s3_stamps_cfg$c.txt 2>&1 || { echo "stamps $c failed"; exit 1; }
done
echo "stamps done"
bash tools/pmc_s3.sh 2 30 && python3 tools/pmc_summary.py gpurun_out/pmc_s3c2 step_kernel > $O/s3_pmc_cfg2.txt || { echo "pmc failed"; exit 1; }
echo "pmc done"
