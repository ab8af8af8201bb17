#!/bin/bash
# Round-4 evidence run (GPU box, from the repo root).  ONE script takes the PMC passes of the three dominant kernels, writes the traffic
# files bench.py's `roofline.traffic` reads (profiles/r04_*_pmc_traffic.json, here on the box and under gpurun_out/r04 for the way back),
# THEN runs the bench command plain and under rocprofv3 --kernel-trace --stats, the d = 256 training step profile and the probe.
# Everything lands under gpurun_out/r04/; tools/publish_r04.py copies what is to be judged into profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
bash tools/pmc_s3.sh 2 30 || { echo "s3 pmc failed"; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_s3c2 step_kernel > $O/s3_pmc_cfg2.txt
python3 tools/pmc_traffic.py gpurun_out/pmc_s3c2 step_kernel $O/r04_s3_f16x3_d32_pmc_traffic.json "eager rollouts of tools/s3_run.py 2 30: B=1000, T=30, n_query=200; one launch = every encoder layer + acquisition logits of ONE design step" > /dev/null || exit 1
echo "s3 pmc done"
bash tools/pmc_cmd.sh x3r4 tools/x3_run.py 30 || { echo "x3 pmc failed"; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_x3r4 "x3::layer_kernel<false>" > $O/x3_pmc.txt
python3 tools/pmc_traffic.py gpurun_out/pmc_x3r4 "x3::layer_kernel<false>" $O/r04_x3_f16x3_d256_pmc_traffic.json "eager rollouts of tools/x3_run.py 30: d=256, F=1024, B=1000, T=30; one launch = one encoder layer of one step" > /dev/null || exit 1
echo "x3 pmc done"
bash tools/pmc_cmd.sh x5r4 tools/d512_run.py 30 || { echo "x5 pmc failed"; exit 1; }
python3 tools/pmc_summary.py gpurun_out/pmc_x5r4 "x5::layer_kernel<false>" > $O/x5_pmc.txt
python3 tools/pmc_traffic.py gpurun_out/pmc_x5r4 "x5::layer_kernel<false>" $O/r04_x5_f16x3_d512_pmc_traffic.json "eager rollouts of tools/d512_run.py 30: d=512, F=128, B=1000, T=30 (the bench's d512 leg); one launch = one encoder layer of one step" > /dev/null || exit 1
echo "x5 pmc done"
cp $O/r04_*_pmc_traffic.json $R/profiles/
cd /tmp && export TMPDIR=/tmp
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || { echo "bench profile failed"; tail -5 $O/bench_prof.err; exit 1; }
cp $(ls -t $O/bench_prof/*/*_kernel_stats.csv | head -1) $R/profiles/r04_bench_kernel_stats.csv
echo "bench profile done"
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_d256_train -- python3 $R/tools/d256_train_run.py > $O/d256_train.log 2>&1 || { echo "d256 train profile failed"; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg3_train -- python3 $R/tools/train_cfg3.py > $O/cfg3_train.log 2>&1 || { echo "cfg3 train profile failed"; exit 1; }
timeout -k 10 300 python3 $R/tools/train_cfg5.py > $O/cfg5_train.log 2>&1 || echo "cfg5 train failed"
timeout -k 10 200 python3 $R/tools/d256_train_time.py > $O/d256_train_time.log 2>&1 || echo "d256 train time failed"
timeout -k 10 200 python3 $R/tools/d256_train_time.py NO_BWD_IMAGE_RECOMPUTE NO_BWD_KV_SPARSE >> $O/d256_train_time.log 2>&1 || echo "d256 train time (switches off) failed"
timeout -k 10 600 python3 $R/tools/config_bench.py > $O/config_bench.jsonl 2> $O/config_bench.err || { echo "config bench failed"; tail -5 $O/config_bench.err; }
$R/tools/probes/mfma_f16x3_ceiling 1.5 > $O/ceiling.log 2>&1
echo "all done"
