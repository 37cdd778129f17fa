#!/bin/bash
out=gpurun_out/r4c; mkdir -p $out
timeout -k 10 420 python -m pytest tests/test_gpu_cells.py tests/test_gpu_worklist.py -m gpu -x -q > $out/t_cells.log 2>&1; echo "cells tests rc=$?"; tail -3 $out/t_cells.log
MF_WAVE_MAX=0 timeout -k 10 420 python -m pytest tests/test_gpu_cells.py -m gpu -x -q > $out/t_cells_team.log 2>&1; echo "cells tests (teams only) rc=$?"; tail -3 $out/t_cells_team.log
timeout -k 10 420 python -m pytest tests/test_gpu_headline.py -m gpu -x -q -k "not room" > $out/t_head.log 2>&1; echo "headline tests rc=$?"; tail -3 $out/t_head.log
tools/dev/kt.sh r4c_def && tools/dev/kt.sh r4c_noteam MF_TEAM_MAX=0 && tools/dev/kt.sh r4c_teamonly MF_WAVE_MAX=0
tools/dev/ab.sh "MF_WAVE_MAX=0 MF_TEAM_MAX=0" "MF_WAVE_MAX=512" "MF_WAVE_MAX=512 --no-pipeline" 2>&1 | tee $out/ab.txt
