#!/bin/bash
# kernel durations of one unpipelined bench run (usage: kt.sh TAG [ENV=..]...)
tag=${1:-kt}; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
for e in "$@"; do export "$e"; done
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$tag -o runc -- python3 bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline --no-pipeline > $out/bench_kt.json 2> $out/kt.err || { tail -5 $out/kt.err; exit 1; }
cp $(find /tmp/kt_$tag -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
grep "mf::" $out/kernel_stats.csv | awk -F'","' '{print $1, $2, $4}' | sed 's/void //; s/(mf::[A-Za-z]*)//' | head -12
