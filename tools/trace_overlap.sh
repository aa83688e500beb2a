#!/bin/bash
# kernel-trace of a short graph-replayed rollout; prints how much the policy / env kernels of the two shards overlap
set -e
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktrace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ktrace -o t -- python3 $ROOT/bench.py --steps 200 --warmup 100 --no-train-loop --no-cpu-baseline $@ > $ROOT/gpurun_out/trace_overlap.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/ktrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fwd_mfma" in r["Kernel_Name"] or "env_kernel" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "P" if "fwd" in r["Kernel_Name"] else "E") for r in rows)
ev = ev[len(ev) // 2:]                      # timed region
t0, t1 = ev[0][0], max(e[1] for e in ev)
busy = sum(e[1] - e[0] for e in ev)
# union length
u, cur_s, cur_e = 0, ev[0][0], ev[0][1]
for s, e, _ in ev[1:]:
    if s > cur_e: u += cur_e - cur_s; cur_s, cur_e = s, e
    else: cur_e = max(cur_e, e)
u += cur_e - cur_s
print(f"kernels {len(ev)}  span {(t1 - t0) / 1e3:.1f} us  sum of durations {busy / 1e3:.1f} us  union {u / 1e3:.1f} us  "
      f"overlap {(busy - u) / 1e3:.1f} us ({100 * (busy - u) / busy:.1f} % of kernel time)  idle {(t1 - t0 - u) / 1e3:.1f} us")
for k in ("P", "E"):
    d = [e[1] - e[0] for e in ev if e[2] == k]
    print(k, "mean duration %.1f us over %d" % (sum(d) / len(d) / 1e3, len(d)))
print("first 12 events (us from t0):", [(round((s - t0) / 1e3, 1), round((e - t0) / 1e3, 1), k) for s, e, k in ev[:12]])
PY
