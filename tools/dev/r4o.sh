#!/bin/bash
out=gpurun_out/r4o; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_reference_shapes.py tests/test_gpu_dropin_agent.py "tests/test_gpu_feature_maps.py::test_two_host_threads_plain_updates_with_different_poses" -m gpu -x -q > $out/t_new.log 2>&1; echo "new tests rc=$?"; tail -12 $out/t_new.log
tools/dev/ab.sh "MF_X=0" "MF_X=0 --no-pipeline" 2>&1 | tee $out/ab.txt
