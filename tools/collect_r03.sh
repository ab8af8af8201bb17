#!/bin/bash
# Round-3 evidence run (GPU box, from the repo root): bench line + its rocprofv3 kernel stats, per-config table, PMC passes of the
# s3 and x3 step / layer kernels, and -- LAST, once -- the PMC command that crashed in round 2 (gpurun_out/pmc_s3_1.log) with the
# process's memory map dumped.  Everything lands under gpurun_out/r03/; what is to be judged is copied into profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || { echo "bench profile failed"; tail -5 $O/bench_prof.err; exit 1; }
echo "bench profile done"
timeout -k 10 600 python3 $R/tools/config_bench.py > $O/config_bench.jsonl 2> $O/config_bench.err || { echo "config bench failed"; tail -5 $O/config_bench.err; exit 1; }
echo "config bench done"
cd $R
bash tools/pmc_s3.sh 2 30 && python3 tools/pmc_summary.py gpurun_out/pmc_s3c2 step_kernel > $O/s3_pmc_cfg2.txt || { echo "s3 pmc failed"; exit 1; }
echo "s3 pmc done"
bash tools/pmc_cmd.sh x3r3 tools/x3_run.py 30 && python3 tools/pmc_summary.py gpurun_out/pmc_x3r3 "layer_kernel<false>" > $O/x3_pmc.txt && python3 tools/pmc_summary.py gpurun_out/pmc_x3r3 "kv_kernel" >> $O/x3_pmc.txt || { echo "x3 pmc failed"; exit 1; }
echo "x3 pmc done"
# the round-2 crash, once (MI355X_MICROARCH / VERDICT r02: do not loop it): same command, memory map dumped just before the timed loop
cd /tmp
ulimit -c 0
ALINE_DUMP_MAPS=$O/pmc_bench_maps.txt timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_bench -- python3 $R/bench.py --precision f16x3 --steps 1 --warmup 1 --graph 0 --no-cpu-baseline --train-steps 0 --no-d256 --no-f32 --no-query-gmm > $O/pmc_bench.log 2>&1
echo "pmc bench.py --graph 0 rc=$?"
tail -3 $O/pmc_bench.log
