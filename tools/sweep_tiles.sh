#!/bin/bash
# Tile-shape / workgroup-size sweep of the fuse pipeline (MF_TILE="s0 s1 s2 threads").
# usage: tools/sweep_tiles.sh out_file "cfg1" "cfg2" ...
out=$1; shift
: > "$out"
for cfg in "$@"; do
  echo "== MF_TILE=$cfg" >> "$out"
  MF_TILE="$cfg" timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'frames/s', {k: round(v,3) for k,v in d['roofline_step']['stage_ms'].items()})" >> "$out" || echo "failed" >> "$out"
done
cat "$out"
