#!/bin/bash
# round-3 session 1: GPU tests, smoke, bench (N = 1 and a 2-rank gloo rehearsal on the one card), one PMC pass of bench.py --graph 0
cd $GRAFT_REPO_ROOT
O=gpurun_out/s1; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" | tee $O/rc.txt
tail -5 $O/gpu_tests.log
timeout -k 10 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $O/rc.txt
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" | tee -a $O/rc.txt
ALINE_BENCH_BACKEND=gloo timeout -k 10 600 python bench.py --gpus 2 --steps 3 --warmup 1 --no-d256 --no-f32 > $O/bench_g2.json 2> $O/bench_g2.err; echo "bench gloo2 rc=$?" | tee -a $O/rc.txt
tail -3 $O/bench_g2.err
