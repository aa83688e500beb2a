#!/bin/bash
# usage: tools/prof_pmc.sh TAG "COUNTER1 COUNTER2 ..."   (one rocprofv3 --pmc pass over a short rollout; keeps only
# the per-kernel averages, written to gpurun_out/pmc_TAG.txt)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
rocprofv3 --pmc $@ --output-format csv -d /tmp/pmc_$TAG -o p -- python3 $ROOT/bench.py --steps 40 --warmup 20 --streams 1 --no-graph --no-train-loop --no-cpu-baseline ${PMC_BENCH_FLAGS} > $ROOT/gpurun_out/pmc_$TAG.log 2>&1
python3 - "$TAG" <<'PY' > $ROOT/gpurun_out/pmc_$TAG.txt
import csv, glob, sys, collections
tag = sys.argv[1]
f = glob.glob(f"/tmp/pmc_{tag}/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    k = (r["Kernel_Name"][:60], r["Counter_Name"])
    acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for (kn, cn), (s, n) in sorted(acc.items()):
    if "fwd_mfma" in kn or "env_kernel" in kn or "rollout_step" in kn or "fwd_h_" in kn or "rollout_w" in kn or "fwd_w_" in kn:
        print(f"{kn:62s} {cn:32s} mean {s / n:14.1f} over {n} dispatches")
PY
cat $ROOT/gpurun_out/pmc_$TAG.txt
