#!/usr/bin/env python3
"""bench.py - headline benchmark of the Com-MARL hot path on MI355X.

Metric (BASELINE.json): env-steps/sec, whole job, PredatorPrey map=10 sen=1 den=0.04 cap=2
(N=M=4), 4096 batched envs per GPU, Comm-DP GNN policy.  One "step" = one pass of the hot
path over the batch: fused policy forward + categorical sample (cm_policy_forward) and the env
step with auto-reset (cm_env_step), both writing straight into the HBM trajectory buffers.
Inputs are resident in HBM when the timed region starts; synthetic data = envs spawned by the
reference's spawn rule on the Philox stream, random-init policy weights (reference init).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line (contract fields + "roofline" + "cpu_baseline" + "train_loop").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

F16_SPLIT_PEAK = 157.3 * 256.0 / 48.0       # TFLOP/s f32-equivalent of the f16-split scheme: 3 x 16 clk against 8 x 32 clk

CONFIGS = {
    # BASELINE.json configs[1] - the configuration the metric is quoted on
    "pp_map10": dict(scenario="pp", map=10, sen=1, n_agents=4, n_preys=4, load=2, max_env_steps=200, loss=0.0,
                     envs=4096, step_us=30, label="PredatorPrey map=10 sen=1 den=0.04 cap=2 (N=M=4), Comm-DP GNN policy"),
    "co_map20": dict(scenario="co", map=20, sen=2, n_agents=24, n_preys=0, load=2, max_env_steps=400, loss=0.0,
                     envs=2048, step_us=100, label="Coverage map=20 sen=2 den=0.06 (N=24), Comm-DP GNN policy"),
    "pp_map30": dict(scenario="pp", map=30, sen=2, n_agents=72, n_preys=72, load=4, max_env_steps=200, loss=0.0,
                     envs=1024, step_us=180, streams=1, label="PredatorPrey map=30 sen=2 den=0.08 cap=4 (N=M=72), Comm-DP GNN policy"),
    "co_map30": dict(scenario="co", map=30, sen=2, n_agents=54, n_preys=0, load=2, max_env_steps=400, loss=0.3,
                     envs=1024, step_us=150, label="Coverage map=30 sen=2 den=0.06 loss=0.3 IID (N=54), Comm-DP GNN policy"),
}


def env_params(c):
    pp = c["scenario"] == "pp"
    return dict(load=c["load"], max_env_steps=c["max_env_steps"], capture_reward=10 if pp else 2,
                step_cost=0.1 if pp else 0, rm=0, penalty=0 if pp else 1, revisit_penalty=0.5, lazy_penalty=1,
                grid_size=c["map"], Rsen=c["sen"], n_agents=c["n_agents"], n_preys=c["n_preys"], n_gcn_layers=2,
                mode="train", trRcom=9, trpl=c["loss"], obstComplex="Easy", add_clock=0)


def algorithmic_bytes(c, d, adj_const, ch_const, L=2):
    """bytes one env-step must move (env kernel, policy kernel)."""
    N, M, pp = c["n_agents"], c["n_preys"], c["scenario"] == "pp"
    G = c["map"] if pp else c["map"] + 2
    adj = 0 if adj_const else 4 * N * N
    ch = 0 if ch_const else 4 * L * N * N
    b_env = 4 * N + 16 * N + ((16 * M + 2 * M) if pp else (2 * (G * G) // 8)) + 8 + 4 * N * d + 5 + adj + ch
    b_pol = 4 * N * d + adj + ch + 4 * N + 20 * N + 4 * N * N
    return b_env, b_pol


def policy_flops(c, d, L=2):
    """FLOPs of one policy forward per env (2 x MACs): encoder, attention, L hops, head."""
    N = c["n_agents"]
    per_agent = 2 * (d * 128 + 128 * 64 + 64 * 64 + N * 64 + L * (64 * 64 + N * 64) + 64 * 128 + 128 * 64 + 64 * 32 + 32 * 5)
    return per_agent * N


def variant_flops(c, d, kind):
    """FLOPs per env of the non-communicating policies."""
    N = c["n_agents"]
    if kind == "obsdp":      # per agent: d -> 128 -> 64 | 64 -> 32 -> 5 (dec_categorical_mlp_policy.py:78-101)
        return 2 * (d * 128 + 128 * 64 + 64 * 32 + 32 * 5) * N
    return 2 * (N * d * 128 + 128 * 64 + 64 * 32 + 32 * 5 * N)      # one chain per env (centralized_...:42-52)


def make_policy(kind, spec, n_agents, device):
    from com_marl_amd import nets
    if kind == "obsdp":
        return nets.DecCategoricalMLPPolicy(spec, n_agents, hidden_sizes=[128, 64, 32], device=device)
    if kind == "cent":
        return nets.CentralizedCategoricalMLPPolicy(spec, n_agents=n_agents, hidden_sizes=[128, 64, 32], device=device)
    return nets.CommCategoricalMLPPolicy(spec, n_agents=n_agents, device=device)


def host_cores():
    """Threads the CPU leg may use: the affinity mask, capped by the cgroup CPU quota and by the
    16-core share a one-GPU box is given."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, int(os.environ.get("COMMARL_CPU_THREADS", "16"))))


def cpu_baseline(c, seed, budget_s=12.0, kind="commdp"):
    """CPU restatement (oracle/, kind 'port') of the same step - C env step + C policy forward +
    sampler, OpenMP over envs on all host cores - timed on a bounded sample of the workload."""
    import numpy as np
    from oracle import oracle as O
    cores = host_cores()
    B = min(c["envs"], 1024)
    cfg = O.make_cfg(c["scenario"], B, c["n_agents"], c["map"], c["sen"], n_preys=c["n_preys"], load=c["load"],
                     max_steps=c["max_env_steps"], channel="IID" if 0 < c["loss"] < 1 else "FC", ploss=c["loss"],
                     seed=seed, rng_mode=O.RNG_PHILOX)
    env = O.OracleEnv(cfg)
    import torch
    from com_marl_amd import envs as E, nets
    spec = E.EnvSpec(E._Box(np.zeros(env.d * env.N), np.ones(env.d * env.N)), E._Discrete(5))
    torch.manual_seed(seed)
    pol = make_policy(kind, spec, env.N, "cpu")                                   # weights only; never run on CPU
    sd = {k: v.detach().numpy() for k, v in pol.state_dict().items()}
    ones = np.ones((B, env.N, 5), np.float32)
    env.reset()

    def one(t):
        if kind == "commdp":
            probs, _ = O.policy_forward(sd, env.obs, ones, env.dist_adj, env.channels, env.N, n_threads=cores)
        elif kind == "obsdp":                           # row-MLP restatement is single-threaded
            probs = O.dec_policy_forward(sd, env.obs.reshape(B, -1), ones, env.N)
        else:
            probs = O.cent_policy_forward(sd, env.obs.reshape(B, -1), ones, env.N)
        env.step(O.sample_actions(probs, seed, 0, t), n_threads=cores)
    one(0)
    one(1)
    t0 = time.perf_counter()
    for t in range(3):
        one(2 + t)
    per = max((time.perf_counter() - t0) / 3, 1e-4)
    n = int(max(3, min(5000, budget_s / per)))
    t0 = time.perf_counter()
    for t in range(n):
        one(5 + t)
    dt = time.perf_counter() - t0
    # the same step on ONE thread (SURVEY.md §8d asks for both ends): a short sample, ~2 s
    def one_st(t):
        if kind == "commdp":
            probs, _ = O.policy_forward(sd, env.obs, ones, env.dist_adj, env.channels, env.N, n_threads=1)
        elif kind == "obsdp":
            probs = O.dec_policy_forward(sd, env.obs.reshape(B, -1), ones, env.N)
        else:
            probs = O.cent_policy_forward(sd, env.obs.reshape(B, -1), ones, env.N)
        env.step(O.sample_actions(probs, seed, 0, t), n_threads=1)
    n1 = int(max(2, min(200, 2.0 / (per * cores))))
    t0 = time.perf_counter()
    for t in range(n1):
        one_st(10000 + t)
    dt1 = time.perf_counter() - t0
    return dict(value=B * n / dt, unit="env-steps/s", cores=cores, kind="port", single_thread_value=B * n1 / dt1,
                sample=f"{n} steps x {B} envs of the same workload (C oracle: env step + policy forward + sample, "
                       f"OpenMP over envs), {dt:.1f} s")


def measure_rollout(config, args, rank, world, dev, steps, warmup, envs=None, streams=None):
    """One config on this rank's GPU: K steps of the rollout under the timing contract (every graph captured, instantiated
    and replayed before t0; barrier + synchronize on both sides; max over ranks), then the per-kernel durations with HIP
    events and the roofline rows.  Returns everything main() puts into the JSON line."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from com_marl_amd import envs as E, nets  # noqa: F401
    from com_marl_amd.rollout import RolloutEngine

    c = dict(CONFIGS[config])
    B = envs or c["envs"]
    if args.scaling == "strong":                        # the batch is split: rank r owns global envs [r*B/k, (r+1)*B/k)
        if B % world:
            raise SystemExit(f"--scaling strong: {B} envs do not split over {world} GPUs")
        B //= world
    c["envs"] = B
    env = E.GridEnvBatch(c["scenario"], env_params(c), B, device=dev, seed=args.seed, env_id_offset=rank * B)
    # env shards per GPU, each with its own policy -> env chain on its own HIP stream.  Two shards pay in steady state
    # (config 2: 36.4 vs 38.6 us/step; configs 4/5 up to +40 %) because the second shard's chain starts ~100 us after the
    # first inside every chunk graph and the two stay out of phase; a timed region of a few ms cannot amortise that
    # start-up stagger (--steps 20 at config 2: 42 us/step as one shard, 45 as two; profiles/r02_steps_sweep.txt), so
    # short runs use one shard.
    # teams of 4 on the wave-owned kernel run a whole chunk as ONE persistent launch (256 workgroups = one wave per SIMD at 4096
    # envs): nothing is left for a second shard to overlap with, so they run as one shard.
    persistent_w = (args.policy == "commdp" and c["n_agents"] == 4 and os.environ.get("COMMARL_PERSISTENT", "1") != "0"
                    and os.environ.get("COMMARL_POLICY_KERNEL", "w")[:1] not in ("h", "f", "v"))
    if streams is not None:
        n_streams = streams
    else:
        n_streams = 1 if (persistent_w or steps * c["step_us"] < 10_000) else c.get("streams", 2)
    ns = n_streams if (n_streams > 1 and B % n_streams == 0) else 1
    if ns > 1:      # the same B envs (same global ids, same Philox streams) as `ns` shards, each on its own stream
        shards = [E.GridEnvBatch(c["scenario"], env_params(c), B // ns, device=dev, seed=args.seed,
                                 env_id_offset=rank * B + k * (B // ns)) for k in range(ns)]
    else:
        shards = [env]
    spec = E.EnvSpec(E._Box(np.zeros(env.d * env.N), np.ones(env.d * env.N)), E._Discrete(5))
    torch.manual_seed(args.seed)                       # replicas: identical weights on every rank
    policy = make_policy(args.policy, spec, env.N, dev)
    policy.set_rng(args.seed, env_id_offset=rank * B)
    G = max(1, min(args.chunk, steps))             # steps per captured hipGraph
    eng = RolloutEngine(shards, policy, horizon=G)        # persistent="auto": one launch per chunk where the wave-owned kernel applies
    eng.reset()
    use_graph = not args.no_graph

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()

    def plan(n):
        """n steps as chunk lengths: whole chunks of G, then one shorter chunk (its own graph) for the remainder."""
        full, rest = divmod(n, G)
        return [G] * full + ([rest] if rest else [])

    def run(n):
        for k in plan(n):
            eng.run_chunk(use_graph=use_graph, n=k, weights_synced=True)   # the weights do not change during a rollout

    # Every graph the timed region replays is captured, instantiated AND replayed once before t0 (capture does not
    # advance the rollout; the warm-up below is W steps through the same chunk machinery, plus one replay of any
    # timed-region graph length the W steps did not already use).
    timed_lengths = sorted(set(plan(steps)))
    if use_graph:
        for k in timed_lengths:
            eng.prepare_graph(k)
    run(warmup)
    extra_warm = [k for k in timed_lengths if k not in set(plan(warmup))]
    for k in extra_warm:
        eng.run_chunk(use_graph=use_graph, n=k)
    n_captured = len(eng._graphs)
    eng.env.check_status()
    policy.sync_weights()
    barrier()
    t0 = time.perf_counter()
    run(steps)
    t_issue = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if os.environ.get("COMMARL_BENCH_DEBUG"):
        print(f"[bench] timed region {dt * 1e6:.1f} us, of which host-side issue {t_issue * 1e6:.1f} us", file=sys.stderr)
    assert len(eng._graphs) == n_captured, "a hipGraph was captured inside the timed region"
    eng.env.check_status()
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = world * B * steps / dt

    # ---- per-kernel durations with HIP events on the launch stream.  The launches are replayed from a
    # captured hipGraph of `inner` back-to-back launches (as in the timed region), so the figure is the
    # GPU-side launch-to-launch period (kernel + ~1.5 us boundary), not Python's call overhead. ----
    def time_kernel(fn, reps=10, inner=20):
        fn()
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        from com_marl_amd import _lib as L
        with L.capture_guard():                     # as RolloutEngine.prepare_graph: no hipFree can land inside the capture
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                for _ in range(inner):
                    fn()
        g.replay()
        torch.cuda.synchronize(dev)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(reps):
            g.replay()
        ev1.record()
        ev1.synchronize()
        return ev0.elapsed_time(ev1) / (reps * inner) * 1e-3                 # seconds per launch

    if getattr(args, "no_kernel_table", False):
        return dict(c=c, B=B, env=env, policy=policy, spec=spec, eng=eng, ns=ns, G=G, value=value, dt=dt, n_captured=n_captured,
                    timed_lengths=timed_lengths, extra_warm=extra_warm, use_graph=use_graph, roofline=None)
    env.reset_all()                                     # full-batch handle: per-launch kernel times at B envs
    adj0 = None if eng.dist_adj is None else eng.dist_adj[0]
    ch0 = None if eng.channels is None else eng.channels[0]
    t_pol = time_kernel(lambda: policy.act_device(
        eng.obs[0].view(B, -1), None, adj0, ch0, out_actions=eng.actions[0], out_probs=eng.probs[0],
        out_attn=None if eng.attn is None else eng.attn[0], policy_step=0, step_base=eng.step_base))
    t_env = time_kernel(lambda: env.step_device(eng.actions[0], out=eng._out(0, 0, B)))
    # the timed region's own launch: policy forward + sample + env step of the whole batch in ONE kernel (cm_rollout_step);
    # None where the library has no fused instantiation for the shape (Obs-DP / CENT policies, unusual obs dims)
    t_fused = None
    if eng._fused_in_graph and hasattr(policy, "step_fused"):
        so = env._out(eng._out(0, 0, B))
        if policy.step_fused(env, eng.obs[0].view(B, -1), adj0, ch0, so, out_actions=eng.actions[0], out_probs=eng.probs[0],
                             out_attn=None if eng.attn is None else eng.attn[0], policy_step=0, step_base=eng.step_base,
                             env_id_offset=rank * B):
            t_fused = time_kernel(lambda: policy.step_fused(
                env, eng.obs[0].view(B, -1), adj0, ch0, so, out_actions=eng.actions[0], out_probs=eng.probs[0],
                out_attn=None if eng.attn is None else eng.attn[0], policy_step=0, step_base=eng.step_base,
                env_id_offset=rank * B))
    # the timed region's launch when the engine runs chunks persistently: ONE cm_rollout_chunk launch = G steps of the whole batch
    t_chunk = None
    if eng._persistent and len(eng.parts) == 1:
        eng.reset()
        if eng.steps_fused(0, G):
            t_chunk = time_kernel(lambda: eng.steps_fused(0, G), reps=5, inner=4)      # seconds per launch of G steps
    env.check_status()
    b_env, b_pol = algorithmic_bytes(c, env.d, env.adj_const, env.ch_const)
    flops = (policy_flops(c, env.d) if args.policy == "commdp" else variant_flops(c, env.d, args.policy)) * B
    if args.policy != "commdp":                         # no mask reads, no attention output
        b_pol = 4 * env.N * env.d + 4 * env.N + 20 * env.N
    kname = "cm_policy_forward" if args.policy == "commdp" else "cm_mlp_policy_forward"
    f16_pipe = args.policy == "commdp" and os.environ.get("COMMARL_POLICY_KERNEL", "h") not in ("f32", "valu")
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and B == CONFIGS[config]["envs"]:      # PMC figures are per launch of the default batch
        try:
            traffic = json.load(open(tpath)).get(config)
        except Exception:
            traffic = None

    def kernel_row(bound, t, fl, nbytes):
        """`bound` names the roofline the kernel is priced on (SURVEY.md §8d names HBM for the path; the policy forward's
        arithmetic intensity - ~700 FLOP/B - puts it under the f32 matrix roof instead); both fractions
        are always emitted: frac against `bound`, hbm_frac = algorithmic bytes / time against 8 TB/s."""
        tf, gbs = fl / t / 1e12, nbytes / t / 1e9
        row = dict(bound=bound, us=t * 1e6, hbm_GBps=gbs, hbm_frac=gbs / 8000.0, algorithmic_bytes=nbytes)
        if bound == "mfma":
            row.update(achieved=tf, peak=157.3, unit="TFLOP/s", frac=tf / 157.3, flops=fl)
            if f16_pipe:                            # the instructions actually issued: 3 f16 MFMAs of 16 clk per 16x16x32 block
                row.update(frac_f16_pipe=tf / F16_SPLIT_PEAK)   # against 8 f32 MFMAs of 32 clk -> ceiling 157.3 x 256 / 48
        else:
            row.update(achieved=gbs, peak=8000.0, unit="GB/s", frac=gbs / 8000.0)
        return row
    kernels = {kname: kernel_row("mfma", t_pol, flops, b_pol * B), "cm_env_step": kernel_row("hbm", t_env, 0, b_env * B)}
    if t_fused is not None:
        kernels["cm_rollout_step"] = kernel_row("mfma", t_fused, flops, (b_env + b_pol) * B)
    if t_chunk is not None:      # per launch: G steps of the whole batch; `us` is the launch, `us_per_step` what one step costs inside it
        kernels["cm_rollout_chunk"] = kernel_row("mfma", t_chunk, flops * G, (b_env + b_pol) * B * G)
        kernels["cm_rollout_chunk"].update(steps_per_launch=G, us_per_step=t_chunk / G * 1e6)
    # the dominant kernel of the TIMED REGION: the persistent chunk, else the fused rollout step when the chunks were captured with it
    dom = "cm_rollout_chunk" if t_chunk is not None else (
        "cm_rollout_step" if t_fused is not None else (kname if t_pol >= t_env else "cm_env_step"))
    tr = (traffic or {}).get(dom) if isinstance(traffic, dict) else None
    if tr is not None and dom == "cm_rollout_chunk":
        tr = tr * G                                     # the file holds bytes per step; a launch runs G steps
    roofline = dict(kernel=dom, **{k: kernels[dom][k] for k in ("bound", "achieved", "peak", "unit", "frac", "hbm_frac")},
                    traffic=tr,
                    traffic_source=(None if tr is None else
                                    "profiles/traffic.json (+ profiles/r03_pmc/): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of a "
                                    "builder run on another MI355X box (not measured by this run), scaled to this launch's step "
                                    "count; raw counter values, the guide's gfx950 x2 FETCH_SIZE rule for 16-byte-per-lane reads NOT "
                                    "applied (the kernel's HBM reads are 4-byte-per-lane state loads)"),
                    note=("peak = dense f32 MFMA (the arithmetic is f32-grade); the instructions issued are three "
                          "v_mfma_f32_16x16x32_f16 per 16x16x32 block on (hi, lo) operand pairs (DESIGN.md §4), whose own ceiling "
                          "is 157.3 x 256/48 = 839 TFLOP/s f32-equivalent: frac_f16_pipe prices the kernel on THAT pipe; "
                          "achieved = algorithmic FLOPs / HIP-event time of graph-replayed back-to-back launches of the whole "
                          "per-GPU batch"),
                    kernels=kernels,
                    hot_path=dict(bound="hbm", achieved=(b_env + b_pol) * B * steps / dt / 1e9 / world, peak=8000.0,
                                  unit="GB/s", frac=(b_env + b_pol) * B * steps / dt / 1e9 / world / 8000.0,
                                  bytes_per_env_step=b_env + b_pol))

    if "frac_f16_pipe" in kernels[dom]:
        roofline["frac_f16_pipe"] = kernels[dom]["frac_f16_pipe"]
    return dict(c=c, B=B, env=env, policy=policy, spec=spec, eng=eng, ns=ns, G=G, value=value, dt=dt, n_captured=n_captured,
                timed_lengths=timed_lengths, extra_warm=extra_warm, use_graph=use_graph, roofline=roofline)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", default="pp_map10", choices=sorted(CONFIGS))
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (default: the config's)")
    ap.add_argument("--chunk", type=int, default=50, help="steps per captured hipGraph")
    ap.add_argument("--streams", type=int, default=None,
                    help="independent env shards per GPU, one HIP stream each (default: the config's, normally 2)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-loop", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true",
                    help="skip the per-kernel HIP-event measurements (profiler runs: every launch of the process is then a launch of "
                         "the timed region's form)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the BASELINE configs 3-5 that a default single-GPU run times after the headline")
    ap.add_argument("--extra-steps", type=int, default=600, help="timed steps of each extra config")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--policy", default="commdp", choices=["commdp", "obsdp", "cent"],
                    help="Comm-DP GNN policy (the headline) or the reference's Obs-DP / CENT variants (SURVEY.md §8f-2)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the config's env batch on EVERY GPU (default); strong: the batch split over the GPUs "
                         "(SURVEY.md §8e: 4096 -> 4096/2048/1024/512 envs per GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start N fresh ranks (one per GPU) and relay rank 0's line.
        # Nothing in this process has touched the GPU yet, and the ranks are children, never a re-exec.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.run(cmd, env=env).returncode)

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU "
                         f"(python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs the MI355X; there is no CPU path")
    # one rank per GPU; COMMARL_DIST_BACKEND=gloo lets several ranks share one GPU to rehearse the N>1 path
    backend = os.environ.get("COMMARL_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"bench.py: {world} ranks but {n_dev} GPU(s) visible - RCCL needs one card per rank "
                         "(COMMARL_DIST_BACKEND=gloo rehearses the N>1 path on fewer cards and says so in its line)")
    rehearsal = world > n_dev                           # several ranks share a card: NOT an N-GPU measurement
    dev = torch.device("cuda", local % n_dev)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    m = measure_rollout(args.config, args, rank, world, dev, args.steps, args.warmup, envs=args.envs, streams=args.streams)
    c, B, env, policy, spec, eng, ns, G, value, dt = (m[k] for k in ("c", "B", "env", "policy", "spec", "eng", "ns", "G", "value", "dt"))
    n_captured, timed_lengths, extra_warm, use_graph, roofline = (m[k] for k in ("n_captured", "timed_lengths", "extra_warm", "use_graph", "roofline"))

    out = {
        # BASELINE.json's metric for the headline config; rollout = fused policy forward + sample + env step + auto-reset
        "metric": ("env-steps/sec (whole node), PredatorPrey M=10 N=4, 4096 envs at 1/2/4/8 GPUs"
                   if (args.config == "pp_map10" and args.policy == "commdp")
                   else f"env-steps/sec (whole node), {args.config}, {args.policy} policy"),
        "value": value, "unit": "env-steps/s", "n_gpus": min(world, n_dev), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (c["label"] if args.policy == "commdp" else c["label"].replace(
            "Comm-DP GNN policy", {"obsdp": "Obs-DP policy (per-agent MLP, no communication)",
                                   "cent": "CENT policy (one MLP over the joint observation)"}[args.policy]))
                               + "; one step = fused policy forward + sample + env step with auto-reset, "
                               "trajectory written to HBM", "envs_per_gpu": B, "total_envs": B * world, "n_agents": c["n_agents"],
                   "obs_dim": env.d, "graph_chunk": 0 if args.no_graph else G, "streams": ns,
                   "step_launches": (f"1 per chunk of {G} steps (cm_rollout_chunk: a wave keeps its envs and the weights for the whole chunk)"
                                     if eng._persistent and eng._fused else
                                     "1 (cm_rollout_step: policy forward + sample + env step fused)" if eng._fused_in_graph
                                     and eng._fused else "2 (cm_policy_forward, cm_env_step)"),
                   "graphs": {"count": n_captured, "chunk_lengths": timed_lengths if use_graph else [],
                              "capture_in_timed_region": False,
                              "warmup_steps_run": args.warmup + sum(extra_warm)},
                   "parallelism": f"env-sharded x{world} (no data-path collective in the rollout)"},
        "roofline": roofline,
    }
    if rehearsal:       # gloo ranks sharing cards: the N>1 code path runs, the number says nothing about N GPUs
        out.update(rehearsal=True, ranks=world, backend=backend)
    # BASELINE configs 3-5 on the same clock (single-GPU default run only): same contract - graphs captured and replayed
    # before t0, synchronize on both sides - with fewer steps; each entry carries its own roofline rows
    if (world == 1 and rank == 0 and args.config == "pp_map10" and args.policy == "commdp" and args.envs is None
            and not args.no_extra_configs and not args.no_graph):
        del m
        extras = {}
        for name in ("co_map20", "pp_map30", "co_map30"):
            try:
                torch.cuda.synchronize(dev)
                x = measure_rollout(name, args, rank, world, dev, args.extra_steps, 100)
                extras[name] = {"workload": x["c"]["label"], "envs_per_gpu": x["B"], "n_agents": x["c"]["n_agents"],
                                "obs_dim": x["env"].d, "steps": args.extra_steps, "warmup": 100, "streams": x["ns"],
                                "value": x["value"], "unit": "env-steps/s", "ms_per_step": x["dt"] / args.extra_steps * 1e3,
                                "step_launches": 1 if (x["eng"]._fused_in_graph and x["eng"]._fused) else 2,
                                "roofline": x["roofline"]}
                del x
            except Exception as e:              # an extra leg never takes the headline line down
                extras[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
        out["configs"] = extras
    if rank == 0 and world == 1 and not args.no_cpu_baseline:     # the CPU leg is a single-GPU-run extra (its OpenMP team
        out["cpu_baseline"] = cpu_baseline(c, args.seed, kind=args.policy)   # would compete with the other ranks' host threads)
    if not args.no_train_loop:
        try:
            from com_marl_amd.train_bench import train_loop_measurement
            out["train_loop"] = train_loop_measurement(env, policy, c, spec, world, rank, dev, args.seed, kind=args.policy)
            # the same loop at the REFERENCE's epoch size (exp_runners/env_uitils.py:13: 60000 N / 8 agent-steps = 30 000 at
            # N = 4, i.e. 7 500 env-steps; the reference collects them with n_envs = 1): 64 envs here, so an epoch is ~120
            # rollout steps + 30 optimiser steps over ~40 paths - the launch-bound end of the update
            if world == 1 and args.policy == "commdp" and args.config == "pp_map10" and args.envs is None:
                from com_marl_amd import envs as E
                small = E.GridEnvBatch(c["scenario"], env_params(c), 64, device=dev, seed=args.seed)
                r = train_loop_measurement(small, policy, c, spec, world, rank, dev, args.seed, epochs=3, kind=args.policy,
                                           batch_size=60000 * c["n_agents"] // 8)
                r["note"] = "reference epoch size: 30 000 agent-steps (7 500 env-steps) per epoch from 64 envs"
                out["train_loop_reference_batch"] = r
        except ImportError:
            out["train_loop"] = None
        except Exception as e:                      # the headline (rollout) line must survive a failure of this extra leg
            out.setdefault("train_loop", {"error": f"{type(e).__name__}: {e}"[:300]})
            if "error" not in (out["train_loop"] or {}):
                out["train_loop_reference_batch"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
