#!/bin/bash
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline $FLAGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$* $FLAGS: frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), {k: round(v,3) for k,v in d['roofline_step']['stage_ms'].items()}, 'parity', d.get('parity',{}).get('within_tolerance'))"; }
FLAGS="--no-pipeline" run MF_X=0
FLAGS="" run MF_X=0
FLAGS="" run MF_CELLS_PER_CU=2
FLAGS="--no-pipeline" run MF_CELLS_PER_CU=2
FLAGS="--no-pipeline" run MF_CELLS_PER_CU=4
FLAGS="--workload room --no-pipeline" run MF_X=0
FLAGS="--workload room" run MF_X=0
