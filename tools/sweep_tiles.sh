#!/bin/bash
# Tile-shape sweep of the fuse pipeline (dev): MF_TILE="s0 s1 s2 threads [gc]" on the 64-frame
# batches (distribution A and B) and on per-frame layer.update() calls.  Run through gpurun.
out=gpurun_out/sweep_tiles.txt
: > $out
for cfg in "" "2 3 3 512" "2 2 3 512" "2 2 3 256" "2 2 2 256" "1 2 3 256"; do
  for wl in distA room; do
    r=$(MF_TILE="$cfg" timeout -k 10 120 python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline_step']['stage_ms']
print(round(d['value']), {k: round(v,3) for k,v in s.items()})")
    echo "batch64 $wl MF_TILE='$cfg' $r" | tee -a $out
  done
done
for cfg in "" "2 2 2 256" "1 2 3 256" "2 2 2 64" "1 1 3 64" "1 2 2 64"; do
  echo "single MF_TILE='$cfg'" | tee -a $out
  MF_TILE="$cfg" timeout -k 10 200 python tools/bench_single.py 24 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('   ', d['workload'], d['map'], 'wall_ms', round(d['ms_per_update_wall'],3), d['gpu_ms'])" | tee -a $out
done
