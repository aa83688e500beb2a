#!/bin/bash
# kernel-trace of the driver's short run (--steps 20 --warmup 5): timeline of the LAST 20 steps' rollout kernels -
# gaps before / between / after them (where the fixed cost of a short timed region sits)
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o t -- python3 $ROOT/bench.py --steps 20 --warmup 5 --no-train-loop --no-cpu-baseline "$@" > $ROOT/gpurun_out/trace_short.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ktrace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the rollout kernels of the timed region are the last 80 (20 steps x 2 shards x 2 kernels) BEFORE the per-kernel timing section
idx = [i for i, r in enumerate(rows) if "fwd_mfma" in r["Kernel_Name"] or "env_kernel" in r["Kernel_Name"]]
# warmup 5 steps -> 20 kernels (+ scratch step 4), then timed 80; find them: after reset kernels
roll = idx[:]
print("rollout-type kernels total", len(roll))
# take kernels number 24+.. heuristically: print a timeline of everything between the first and the 110th rollout kernel
lo, hi = roll[0], roll[min(len(roll) - 1, 130)]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = t0
for r in rows[lo:hi + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"]
    tag = "P" if "fwd_mfma" in name else ("E" if "env_kernel" in name else name[:40])
    print(f"{(s - t0) / 1e3:9.1f} {(e - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:6.1f}  gap_from_prev_end {(s - prev_end) / 1e3:7.1f}  q{r.get('Queue_Id','?')} {tag}")
    prev_end = max(prev_end, e)
PY
