#!/bin/bash
# dev loop on the GPU box: cells tests + headline bench + stamps; usage: tools/dev/quick3.sh TAG [variant-lib ...]
tag=${1:-q}; shift
mkdir -p gpurun_out/$tag
timeout -k 10 600 python -m pytest tests/test_gpu_cells.py tests/test_gpu_dense_scene.py tests/test_gpu_splat.py tests/test_gpu_edge.py -m gpu -x -q > gpurun_out/$tag/pytest.log 2>&1; echo "pytest rc $?"; tail -4 gpurun_out/$tag/pytest.log
for lib in "" "$@"; do
  name=${lib:-default}; name=$(basename $name .so)
  MASSFUSE_LIB=$lib timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-extras --cpu-frames 1 > gpurun_out/$tag/b_$name.json 2> gpurun_out/$tag/b_$name.err || tail -3 gpurun_out/$tag/b_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/$tag/b_$name.json"))
print("$name", "frames/s", round(d["value"]), "frac", round(d["roofline"]["frac"],3), "alone_ms", d["roofline"]["kernel_ms_unoverlapped"], {k: round(v,3) for k,v in d["roofline_step"]["stage_ms"].items()}, "parity", d.get("parity",{}).get("within_tolerance"), d.get("parity",{}).get("occupancy_bit_exact"))
PY
  MASSFUSE_LIB=$lib MF_STAMPS=1 timeout -k 10 200 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-pipeline 2>&1 >/dev/null | grep "STAMPS. cells kernel:" | tail -1
done
