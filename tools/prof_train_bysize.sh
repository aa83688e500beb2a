#!/bin/bash
# rocprofv3 kernel trace of the default bench, summarised per (kernel, grid size): separates the training launches of a
# kernel from its full-batch / rollout launches.  Output: gpurun_out/prof_train_bysize.txt
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_bysize -o train -- python3 $ROOT/bench.py --steps 200 --warmup 50 --no-cpu-baseline --no-extra-configs --no-kernel-table > $ROOT/gpurun_out/prof_train_bysize.log 2>&1
python3 - <<'PY' > $ROOT/gpurun_out/prof_train_bysize.txt
import csv, glob, collections
f = glob.glob('/tmp/prof_bysize/**/*kernel_trace.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(f)):
    k = (r['Kernel_Name'][:70], int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r['Grid_Size']))
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    acc[k][0] += 1; acc[k][1] += d
tot = sum(v[1] for v in acc.values())
print('total kernel s', tot / 1e9)
for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{k[0]:70s} grid {k[1]:>10d} calls {v[0]:6d} total {v[1] / 1e6:9.2f} ms avg {v[1] / v[0] / 1e3:9.1f} us")
PY
tail -1 $ROOT/gpurun_out/prof_train_bysize.log | cut -c1-200
