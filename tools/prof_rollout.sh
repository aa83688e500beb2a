#!/bin/bash
# usage: tools/prof_rollout.sh TAG [CONFIG [STEPS [extra bench.py flags]]] - rocprofv3 kernel-trace summary of a rollout
# bench (no train loop); keeps only the stats csv.  With `--streams 1` every launch covers the whole per-GPU batch, so
# the kernel averages are directly comparable with bench.py's roofline rows.
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-rollout}
CFG=${2:-pp_map10}
STEPS=${3:-2000}
shift; shift; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -o $TAG -- python3 $ROOT/bench.py --config $CFG --steps $STEPS --warmup 100 --no-train-loop --no-cpu-baseline "$@" > $ROOT/gpurun_out/prof_$TAG.log 2>&1
find $ROOT/gpurun_out/prof_$TAG -type f ! -name '*stats*.csv' -delete
grep -v "^[WE]2026" $ROOT/gpurun_out/prof_$TAG.log | tail -1 | cut -c1-300
