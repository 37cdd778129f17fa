#!/bin/bash
# kernel durations + SQ counters of the tile kernels, unpipelined (usage: r4b.sh TAG [ENV=..])
tag=${1:-r4b}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for e in "$@"; do export "$e"; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o runc -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-pipeline > $out/bench_kt.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
cp $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
head -12 $out/kernel_stats.csv
tools/dev/sq.sh $tag
