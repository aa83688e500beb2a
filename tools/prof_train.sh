#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench (rollout + train loop); output under gpurun_out/prof_train
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_train -o train -- python3 $ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extra-configs --no-kernel-table > $ROOT/gpurun_out/prof_train.log 2>&1
find $ROOT/gpurun_out/prof_train -type f ! -name '*stats*.csv' -delete
du -sh $ROOT/gpurun_out
tail -2 $ROOT/gpurun_out/prof_train.log
