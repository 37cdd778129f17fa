#!/bin/bash
out=gpurun_out/r4e; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_cells.py -m gpu -x -q > $out/t_cells.log 2>&1; echo "cells tests rc=$?"; tail -5 $out/t_cells.log
timeout -k 10 300 python -m pytest tests/test_gpu_dense_scene.py tests/test_gpu_splat.py tests/test_gpu_edge.py -m gpu -x -q > $out/t_dense.log 2>&1; echo "dense/splat/edge tests rc=$?"; tail -5 $out/t_dense.log
tools/dev/kt.sh r4e_kt
tools/dev/ab.sh "" "--no-pipeline" "--workload room" 2>&1 | tee $out/ab.txt
