#!/bin/bash
tag=${1:-r4g}
out=gpurun_out/$tag; mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_gpu_cells.py -m gpu -x -q > $out/t_cells.log 2>&1; echo "cells tests rc=$?"; tail -3 $out/t_cells.log
tools/dev/sq.sh $tag > /dev/null; grep "fuse_cells\|scatter\|count_k" gpurun_out/$tag/sq_summary.txt
for a in "" "--no-pipeline"; do timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline $a 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$a frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), 'alone', d['roofline']['kernel_ms_unoverlapped'] and round(d['roofline']['kernel_ms_unoverlapped'],3), {k: round(v,3) for k,v in d['roofline_step']['stage_ms'].items()})"; done
