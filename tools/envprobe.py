"""In-kernel phase clocks of the env step (COMMARL_ENV_STOP=-1 -> ENV_PROBE stamps of workgroup 0, cm_env_dev.h).
usage: python tools/envprobe.py [config] [envs]   - eager steps only (the launch synchronises to print)"""
import os, sys
os.environ["COMMARL_ENV_STOP"] = "-1"
sys.path.insert(0, '.')
import torch
from com_marl_amd import envs as E
import bench
cfg = sys.argv[1] if len(sys.argv) > 1 else "pp_map10"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
c = dict(bench.CONFIGS[cfg])
env = E.GridEnvBatch(c["scenario"], bench.env_params(c), B, device="cuda:0", seed=1)
env.reset_all()
for i in range(8):
    act = torch.randint(0, 5, (B, env.N), dtype=torch.int32, device="cuda:0")
    env.step_device(act)
env.check_status()
