#!/bin/bash
# round 4, first GPU check of fuse_wave_kernel: parity tests, then A/B of the headline with the kernel off / on
out=gpurun_out/r4a; mkdir -p $out
timeout -k 10 420 python -m pytest tests/test_gpu_cells.py tests/test_gpu_worklist.py -m gpu -x -q > $out/t_cells.log 2>&1; echo "cells tests rc=$?"; tail -3 $out/t_cells.log
timeout -k 10 420 python -m pytest tests/test_gpu_headline.py -m gpu -x -q > $out/t_head.log 2>&1; echo "headline tests rc=$?"; tail -3 $out/t_head.log
tools/dev/ab.sh "MF_WAVE_MAX=0" "MF_WAVE_MAX=512" "MF_WAVE_MAX=256" "MF_WAVE_MAX=128" "MF_WAVE_MAX=0 --no-pipeline" "MF_WAVE_MAX=512 --no-pipeline" 2>&1 | tee $out/ab.txt
