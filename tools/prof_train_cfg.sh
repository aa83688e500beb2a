#!/bin/bash
# rocprofv3 kernel-trace summary of the train loop at another config: tools/prof_train_cfg.sh co_map20
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
CFG=${1:-co_map20}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_train_$CFG -o train -- python3 $ROOT/bench.py --config $CFG --steps 100 --warmup 50 --no-cpu-baseline > $ROOT/gpurun_out/prof_train_$CFG.log 2>&1
find $ROOT/gpurun_out/prof_train_$CFG -type f ! -name '*kernel_stats*.csv' -delete
tail -1 $ROOT/gpurun_out/prof_train_$CFG.log | cut -c1-100
