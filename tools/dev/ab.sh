#!/bin/bash
# dev: A/B of bench configurations: tools/dev/ab.sh "ENV=... [--flag]" ...   (each item: env assignments then bench flags)
for cfg in "$@"; do
  envs=""; flags=""
  for w in $cfg; do case $w in --*) flags="$flags $w";; *=*) envs="$envs $w";; *) flags="$flags $w";; esac; done
  echo "== $cfg"
  env $envs timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline $flags 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   frames/s', round(d['value']), 'ms/step', round(d['ms_per_step'],3), 'frac', round(d['roofline']['frac'],3), 'alone', d['roofline']['kernel_ms_unoverlapped'] and round(d['roofline']['kernel_ms_unoverlapped'],3), {k: round(v,3) for k,v in d['roofline_step']['stage_ms'].items()})"
done
