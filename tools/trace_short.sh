#!/bin/bash
# kernel trace of the driver's short run (--steps 20 --warmup 5): groups of back-to-back fused-step launches (a group = one
# graph replay), their span, summed kernel time and the gaps between launches - where a short timed region's time goes
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o t -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-train-loop --no-cpu-baseline "$@" > $ROOT/gpurun_out/trace_short.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ktrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
groups, cur = [], []
for r in rows:
    if "rollout_step_kernel" not in r["Kernel_Name"] and "chunk_tail" not in r["Kernel_Name"]:
        if cur: groups.append(cur); cur = []
        continue
    s = int(r["Start_Timestamp"])
    if cur and s - int(cur[-1]["End_Timestamp"]) > 20000:
        groups.append(cur); cur = []
    cur.append(r)
if cur: groups.append(cur)
for g in groups[:8]:
    ks = [r for r in g if "rollout_step" in r["Kernel_Name"]]
    if len(ks) < 5: continue
    span = (int(g[-1]["End_Timestamp"]) - int(g[0]["Start_Timestamp"])) / 1e3
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in ks]
    gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(g[:-1], g[1:])]
    print(f"group of {len(ks):3d} steps: span {span:7.1f} us, kernel time {sum(durs):7.1f} us (mean {sum(durs)/len(durs):5.2f}, min {min(durs):5.2f}, max {max(durs):5.2f}), "
          f"gaps mean {sum(gaps)/max(len(gaps),1):4.2f} max {max(gaps) if gaps else 0:4.2f} us")
PY
