timeout -k 10 200 python tools/bench_single.py 24 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    print('   ', d['workload'], d['map'], 'wall_ms', round(d['ms_per_update_wall'],3), 'fuse', d['gpu_ms']['fuse_tiles'], 'call', d['gpu_ms']['call'])"
timeout -k 10 600 python -m pytest tests/test_gpu_splat.py tests/test_gpu_edge.py tests/test_gpu_fullsize.py tests/test_gpu_episode.py tests/test_gpu_headline.py -m gpu -x -q -k "not headline_launch_64 and not two_ranks" 2>&1 | tail -3
