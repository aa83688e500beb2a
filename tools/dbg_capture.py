"""Capture status after every operation of a RolloutEngine chunk capture (hipStreamIsCapturing on every stream): which call
invalidates a capture.  python tools/dbg_capture.py 1,3   (shard counts, run in sequence in one process)"""
import ctypes, sys
sys.path.insert(0, '.')
import numpy as np, torch
from com_marl_amd import envs as E, nets, rollout
from com_marl_amd.rollout import RolloutEngine
hip = ctypes.CDLL("libamdhip64.so")
def status(st):
    s = ctypes.c_int(-1)
    rc = hip.hipStreamIsCapturing(ctypes.c_void_p(st.cuda_stream), ctypes.byref(s))
    return rc, s.value
params = dict(load=2, max_env_steps=9, capture_reward=10, step_cost=0.1, rm=0, penalty=0, grid_size=10, Rsen=1,
              n_agents=4, n_preys=4, n_gcn_layers=2, mode="train", trRcom=9, trpl=0.4)
for shards in [int(x) for x in sys.argv[1].split(',')]:
    print('=== shards', shards, flush=True)
    B = 96
    es = [E.GridEnvBatch("pp", params, B // shards, device="cuda:0", seed=4, env_id_offset=1000 + k * (B // shards)) for k in range(shards)]
    spec = E.EnvSpec(E._Box(np.zeros(84), np.ones(84)), E._Discrete(5))
    torch.manual_seed(2)
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=4, device="cuda:0")
    pol.set_rng(4, env_id_offset=1000)
    eng = RolloutEngine(es, pol, horizon=6)
    eng.reset()
    orig_sp, orig_tail, orig_fork, orig_join = eng._step_part, eng._chunk_tail, eng.fork, eng.join
    def chk(tag):
        if eng._capturing:
            cur = torch.cuda.current_stream()
            print(tag, "cur", status(cur), [status(s) for s in eng.streams if s is not None], flush=True)
    def sp(k, t, g):
        try:
            orig_sp(k, t, g)
        finally:
            chk(f"step_part k={k} t={t}")
    def tail(k, n):
        orig_tail(k, n); chk(f"tail k={k}")
    def fork():
        orig_fork(); chk("fork")
    def join():
        orig_join(); chk("join")
    eng._step_part, eng._chunk_tail, eng.fork, eng.join = sp, tail, fork, join
    eng.run_chunk(use_graph=True)
    torch.cuda.synchronize()
    print("ok")
