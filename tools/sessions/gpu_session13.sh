#!/bin/bash
# session 13: in-kernel phase stamps of x5::layer_kernel (cfg5 geometry B = 256, 203 rows)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s13; mkdir -p $O
ALINE_HIP_LIB=$PWD/aline_amd/csrc/variants/lib_stamps.so timeout -k 10 200 python tools/x3_stamps.py 30 512 128 256 > $O/x5_stamps_F128.txt 2>&1; echo "stamps rc=$?"; tail -6 $O/x5_stamps_F128.txt
ALINE_HIP_LIB=$PWD/aline_amd/csrc/variants/lib_stamps.so timeout -k 10 200 python tools/x3_stamps.py 30 512 2048 256 > $O/x5_stamps_F2048.txt 2>&1; echo "stamps rc=$?"; tail -6 $O/x5_stamps_F2048.txt
