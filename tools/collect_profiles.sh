#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): kernel trace + stats of the bench
# command, then FETCH_SIZE / WRITE_SIZE / L2 hit counters in separate passes (the guide's recipe),
# then the per-frame update path.  Summaries: python tools/summarize_profiles.py <tag>  (afterwards, here).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof
rm -rf $out && mkdir -p $out
BENCH="python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o runc -- $BENCH > $out/bench_under_kt.json 2> $out/kt.err || exit 1
echo "kernel trace done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/fetch -o runc -- python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> $out/fetch.err || exit 1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/write -o runc -- python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> $out/write.err || exit 1
echo "write done"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $out/tcc -o runc -- python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline > /dev/null 2> $out/tcc.err || exit 1
echo "tcc done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_single -o runc -- python3 tools/bench_single.py 48 > $out/single_frame_updates.jsonl 2> $out/kt_single.err || exit 1
echo "single-frame trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_maps -o runc -- python3 tools/bench_maps.py 120 shared > $out/shared_frame_under_kt.jsonl 2> $out/kt_maps.err || exit 1
python3 tools/bench_maps.py 300 > $out/shared_frame_updates.jsonl 2>> $out/kt_maps.err || exit 1
echo "three-map step done"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/sq -o runc -- python3 bench.py --steps 4 --warmup 1 --no-extras --no-cpu-baseline --no-pipeline > /dev/null 2> $out/sq.err || exit 1
python3 tools/pmc_summary.py $out/sq mf:: > $out/sq_summary.txt
echo "sq done"
# distribution B (box-room trajectory): the all-integer tile kernel; kernel trace + HBM counters
BENCHB="python3 bench.py --workload room --steps 20 --warmup 5 --no-extras --no-cpu-baseline"
outb=gpurun_out/prof_room
rm -rf $outb && mkdir -p $outb
rocprofv3 --kernel-trace --stats --output-format csv -d $outb/kt -o runc -- $BENCHB > $outb/bench_under_kt.json 2> $outb/kt.err || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $outb/fetch -o runc -- python3 bench.py --workload room --steps 6 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2> $outb/fetch.err || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $outb/write -o runc -- python3 bench.py --workload room --steps 6 --warmup 3 --no-extras --no-cpu-baseline > /dev/null 2> $outb/write.err || exit 1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $outb/sq -o runc -- python3 bench.py --workload room --steps 6 --warmup 3 --no-extras --no-cpu-baseline --no-pipeline > /dev/null 2> $outb/sq.err || exit 1
python3 tools/pmc_summary.py $outb/sq mf:: > $outb/sq_summary.txt
echo "room done"
# summarise here (the raw traces are too bulky to travel back), keep the summaries only
mkdir -p gpurun_out/prof_summary
MF_PROFILE_KERNEL='mf::fuse_cells_kernel<1, 7, false, false, 8>' MF_PROFILE_OUT=gpurun_out/prof_summary python3 tools/summarize_profiles.py ${1:-r03} distA_sequential_b64 > gpurun_out/prof_summary/summary.log 2>&1 || { tail -5 gpurun_out/prof_summary/summary.log; exit 1; }
MF_PROFILE_SRC=$outb MF_PROFILE_KERNEL='mf::fuse_cells_kernel<1, 7, false, true, 4>' MF_PROFILE_OUT=gpurun_out/prof_summary python3 tools/summarize_profiles.py ${1:-r03}_room room_sequential_b64 > gpurun_out/prof_summary/summary_room.log 2>&1 || { tail -5 gpurun_out/prof_summary/summary_room.log; exit 1; }
cp $outb/bench_under_kt.json gpurun_out/prof_summary/${1:-r03}_room_bench_under_rocprof.json
cp $outb/sq_summary.txt gpurun_out/prof_summary/${1:-r03}_room_sq_counters.txt
rm -rf $outb
cp $out/bench_under_kt.json gpurun_out/prof_summary/${1:-r03}_bench_under_rocprof.json
cp $out/single_frame_updates.jsonl gpurun_out/prof_summary/${1:-r03}_single_frame_updates.jsonl
cp $out/shared_frame_updates.jsonl gpurun_out/prof_summary/${1:-r03}_three_map_step.jsonl
cp $out/kt_maps/runc/*_kernel_stats.csv gpurun_out/prof_summary/${1:-r03}_three_map_step_kernel_stats.csv 2>/dev/null || cp $(find $out/kt_maps -name "*kernel_stats.csv" | head -1) gpurun_out/prof_summary/${1:-r03}_three_map_step_kernel_stats.csv
cp $out/sq_summary.txt gpurun_out/prof_summary/${1:-r03}_sq_counters.txt
rm -rf $out
du -sh gpurun_out/prof_summary; ls gpurun_out/prof_summary
