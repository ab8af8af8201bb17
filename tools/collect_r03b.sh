#!/bin/bash
# Round-3 evidence run on the final build (GPU box, from the repo root): bench line + rocprofv3 kernel stats of the same command,
# per-config table, kernel stats of cfg5 (x5) and of the training steps, PMC passes of the s3 / x3 / x5 dominant kernels.
# Everything lands under gpurun_out/r03b/; what is to be judged is copied into profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03b; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || { echo "bench profile failed"; tail -5 $O/bench_prof.err; exit 1; }
echo "bench profile done"
timeout -k 10 600 python3 $R/tools/config_bench.py > $O/config_bench.jsonl 2> $O/config_bench.err || { echo "config bench failed"; tail -5 $O/config_bench.err; exit 1; }
echo "config bench done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/tools/cfg5_run.py f16x3 128 > $O/cfg5_prof.log 2>&1 || { echo "cfg5 profile failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3_train -- python3 $R/tools/train_cfg3.py > $O/cfg3_train.log 2>&1 || { echo "cfg3 train profile failed"; exit 1; }
echo "config profiles done"
cd $R
bash tools/pmc_s3.sh 2 30 && python3 tools/pmc_summary.py gpurun_out/pmc_s3c2 step_kernel > $O/s3_pmc_cfg2.txt || { echo "s3 pmc failed"; exit 1; }
echo "s3 pmc done"
bash tools/pmc_cmd.sh x3r3 tools/x3_run.py 30 && python3 tools/pmc_summary.py gpurun_out/pmc_x3r3 "x3::layer_kernel<false>" > $O/x3_pmc.txt || { echo "x3 pmc failed"; exit 1; }
echo "x3 pmc done"
bash tools/pmc_cmd.sh x5r3 tools/cfg5_run.py f16x3 128 && python3 tools/pmc_summary.py gpurun_out/pmc_x5r3 "x5::layer_kernel<false>" > $O/x5_pmc.txt || { echo "x5 pmc failed"; exit 1; }
echo "x5 pmc done"
