#!/bin/bash
# quick correctness + speed check of the fuse pipeline on the GPU box (dev loop)
# usage: tools/quick_gpu.sh [tag]   (run through gpurun)
tag=${1:-q}
mkdir -p gpurun_out/$tag
timeout -k 10 300 python -m pytest tests/test_gpu_splat.py tests/test_gpu_edge.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-extras --cpu-frames 1 > gpurun_out/$tag/b.json 2> gpurun_out/$tag/b.err || tail -5 gpurun_out/$tag/b.err
python - <<PY
import json
d=json.load(open("gpurun_out/$tag/b.json"))
print("frames/s", round(d["value"]), "frac", round(d["roofline"]["frac"],3), {k: round(v,3) for k,v in d["roofline_step"]["stage_ms"].items()}, "parity", d["parity"]["within_tolerance"], d["parity"]["occupancy_bit_exact"], d["parity"]["max_scaled_err"])
PY
MF_STAMPS=1 timeout -k 10 120 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline 2>&1 | grep STAMPS | tail -1
