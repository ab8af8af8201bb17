#!/bin/bash
# Round-2 evidence for the training step (GPU box, repo root): bench line, kernel stats of the bench and of a
# profiled training probe (t_chunk 30), PMC utilisation counters and HBM traffic of the backward kernels.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02t; rm -rf $O $R/gpurun_out/pmc_train2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || { echo "bench failed"; tail -5 $O/bench.err; exit 1; }
echo "bench done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_prof -- python3 $R/bench.py --no-cpu-baseline > $O/bench_prof.json 2> $O/bench_prof.err || { echo "bench profile failed"; exit 1; }
echo "bench profile done"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/train_prof -- python3 $R/tools/train_probe.py 30 > $O/train_prof.log 2>&1 || { echo "train profile failed"; exit 1; }
grep t_chunk $O/train_prof.log
cd $R
bash tools/pmc_cmd.sh train2 tools/train_probe.py 30 || exit 1
for k in "tail_kernel<true" attn_block_bwd layer_fwd acqb::bwd acqb::logit gmm_bwd; do python3 tools/pmc_summary.py gpurun_out/pmc_train2 "$k"; done > $O/train_pmc_summary.txt
echo "pmc done"
