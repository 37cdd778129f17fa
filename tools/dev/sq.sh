#!/bin/bash
# dev: SQ instruction / wait counters of the tile kernels for one bench configuration
# usage (through gpurun): tools/dev/sq.sh TAG [bench args...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/$tag; mkdir -p $out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/sq_$tag -o runc -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-pipeline "$@" > /dev/null 2> $out/sq.err || { tail -5 $out/sq.err; exit 1; }
python3 tools/pmc_summary.py /tmp/sq_$tag mf:: > $out/sq_summary.txt
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d /tmp/sq2_$tag -o runc -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-pipeline "$@" > /dev/null 2> $out/sq2.err || { tail -5 $out/sq2.err; exit 1; }
python3 tools/pmc_summary.py /tmp/sq2_$tag mf:: >> $out/sq_summary.txt
cat $out/sq_summary.txt
