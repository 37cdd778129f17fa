for wl in distA room; do
  timeout -k 10 120 python bench.py --workload $wl --steps 10 --warmup 2 --no-extras --cpu-frames 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['roofline_step']['stage_ms']
print('$wl', round(d['value']), {k: round(v,3) for k,v in s.items()}, d['parity']['within_tolerance'])"
done
