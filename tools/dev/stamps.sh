#!/bin/bash
# dev: phase stamps of the tile kernel for a few configurations: tools/dev/stamps.sh "ENV=..." "ENV=..."
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg MF_STAMPS=1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-pipeline 2>&1 >/dev/null | grep "STAMPS" | grep -v "^\[MF_STAMPS\] blocks" | tail -1
  env $cfg timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --no-pipeline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   frames/s', round(d['value']), {k: round(v,3) for k,v in d['roofline_step']['stage_ms'].items()})"
done
