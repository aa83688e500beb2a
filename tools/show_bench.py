"""One-screen summary of a bench.py JSON line:  python tools/show_bench.py FILE"""
import json
import sys

d = json.loads(open(sys.argv[1]).readlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "steps", "warmup", "n_gpus")})
r = d.get("roofline") or {}
if r:
    ch = (r.get("kernels") or {}).get(r.get("kernel"), {})
    print("roofline:", r.get("kernel"), "frac", round(r["frac"], 4), "f16 pipe", round(r.get("frac_f16_pipe", 0), 4), "hbm_frac",
          round(r.get("hbm_frac", 0), 4), "us", round(ch.get("us", 0), 2), "us_per_step", round(ch.get("us_per_step", 0), 3), "traffic", r.get("traffic"))
for k in ("train_loop", "train_loop_reference_batch"):
    t = d.get(k)
    if isinstance(t, dict) and "value" in t:
        print(k + ":", round(t["value"]), "env-steps/s, rollout", round(t["rollout_s_per_epoch"], 5), "s, update", round(t["update_s_per_epoch"], 5), "s")
for k, v in (d.get("configs") or {}).items():
    if "value" in v:
        print("config", k + ":", round(v["value"]), "env-steps/s,", round(v["ms_per_step"], 4), "ms/step")
c = d.get("cpu_baseline")
if c:
    print("cpu_baseline:", round(c["value"]), c["unit"], "on", c["cores"], "cores")
