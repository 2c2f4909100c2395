#!/bin/bash
# Kernel timeline of overlapped steps (rocprofv3 --kernel-trace only): start / end / queue of every dispatch, for
# tools/experiments/timeline_gaps.py.   usage (GPU box): tools/experiments/timeline.sh <cfg> [extra bench args]
CFG=${1:-cfg2}; shift
OUT=$PWD/gpurun_out/timeline_$CFG
mkdir -p $OUT
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python $ROOT/bench.py --config $CFG --steps 3 --warmup 2 --steps-only --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err
cd $ROOT
find $OUT -name "*kernel_trace.csv" -exec mv {} $OUT/kernel_trace.csv \;
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
ls -la $OUT
