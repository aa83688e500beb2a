"""The two non-default net options the reference CLI reaches (exp_runners/env_uitils.py:85,88): attention_type='dot'
(com_marl/torch/modules/attention_module.py:38-41) and the critic's aggregator_type='direct'
(com_marl/torch/baselines/comm_base_critic.py:48-49,84-87,115-118), against recordings of the reference classes
(tests/golden/net_options_*.npz, oracle/gen_golden.py::record_net_options): forward outputs through the oracle (CPU) and
through the HIP kernels (GPU: fused rollout forward, training path), and every parameter gradient of a PPO-shaped scalar /
the Gaussian NLL through the training path.  Tolerances: 1e-5 (north_star's float bar); gradients 1e-4 relative + 1e-5 of
the tensor's scale, as tests/test_hip_ppo_parity.py."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_golden import GOLDEN

FIXTURES = {"net_options_pp_map10": 4, "net_options_co_map20": 24}


def _sd(z, tag, pre):
    k0 = f"{tag}.{pre}."
    return {k[len(k0):]: z[k] for k in z.files if k.startswith(k0)}


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_oracle_matches_reference_net_options(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N = FIXTURES[name]
    obs, adj, ch = z["obs"], z["adj"], z["channels"]
    S = obs.shape[0]
    for tag in ("dot", "direct"):
        pol, crit = _sd(z, tag, "pol"), _sd(z, tag, "crit")
        assert ("attention_layer.linear_in.weight" in pol) == (tag == "direct")        # 'dot' has no linear_in at all
        probs, attn = O.policy_forward(pol, obs, np.ones((S, N, 5), np.float32), adj, ch, N)
        np.testing.assert_allclose(probs, z[f"{tag}.probs"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(attn, z[f"{tag}.attn"], rtol=1e-5, atol=1e-5)
        values = O.critic_forward(crit, obs, adj, ch, N)
        np.testing.assert_allclose(values, z[f"{tag}.values"], rtol=1e-5, atol=2e-5)
        np.testing.assert_allclose(O.critic_loss(values, z["returns"]), z[f"{tag}.critic_loss"], rtol=1e-5, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_hip_net_options_forward_and_training_path(name):
    import torch
    from torch.distributions import Categorical
    from com_marl_amd import envs as E, nets
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    N = FIXTURES[name]
    dev = "cuda:0"
    obs, adj, ch = (torch.as_tensor(z[k]).to(dev) for k in ("obs", "adj", "channels"))
    S, d = obs.shape[0], obs.shape[1] // N
    spec = E.EnvSpec(E._Box(np.zeros(d * N), np.ones(d * N)), E._Discrete(5))
    acts = torch.as_tensor(z["actions"]).to(dev)
    wts, returns = torch.as_tensor(z["weights"]).to(dev), torch.as_tensor(z["returns"]).to(dev)
    for tag, att, agg in (("dot", "dot", "sum"), ("direct", "general", "direct")):
        pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, attention_type=att, device=dev)
        crit = nets.CommBaseCritic(spec, n_agents=N, attention_type=att, aggregator_type=agg, device=dev)
        ref_pol, ref_crit = _sd(z, tag, "pol"), _sd(z, tag, "crit")
        assert set(pol.state_dict()) == set(ref_pol) and set(crit.state_dict()) == set(ref_crit)     # checkpoint interchange
        pol.load_state_dict({k: torch.as_tensor(v) for k, v in ref_pol.items()})
        crit.load_state_dict({k: torch.as_tensor(v) for k, v in ref_crit.items()})
        # rollout forward (fused kernel) and no-grad critic
        _, probs, attn = pol.act_device(obs, None, adj, ch, want_actions=False, policy_step=0)
        np.testing.assert_allclose(probs.cpu().numpy(), z[f"{tag}.probs"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(attn.cpu().numpy(), z[f"{tag}.attn"], rtol=1e-5, atol=1e-5)
        with torch.no_grad():
            v = crit.forward(obs, None, adj, ch)
        np.testing.assert_allclose(v.cpu().numpy(), z[f"{tag}.values"], rtol=1e-5, atol=2e-5)
        # training path: the recorded scalar and every gradient
        p_train, _ = pol._probs(obs, None, adj, ch)
        dist = Categorical(probs=p_train)
        scalar = -(dist.log_prob(acts).sum(-1) * wts).mean() - 0.1 * dist.entropy().mean(-1).mean()
        pol.zero_grad()
        scalar.backward()
        loss = crit.compute_loss(obs, returns, adj, ch)
        crit.zero_grad()
        loss.backward()
        np.testing.assert_allclose(scalar.item(), float(z[f"{tag}.scalar"]), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(loss.item(), float(z[f"{tag}.critic_loss"]), rtol=1e-5, atol=1e-5)
        for pre, net in (("gpol", pol), ("gcrit", crit)):
            for pname, p in net.named_parameters():
                want = z[f"{tag}.{pre}.{pname}"]
                got = np.zeros_like(want) if p.grad is None else p.grad.cpu().numpy()
                np.testing.assert_allclose(got, want, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(want).max())), err_msg=f"{tag} {pre} {pname}")


def test_unbuilt_options_say_so():
    from com_marl_amd import nets
    with pytest.raises(NotImplementedError, match="'diff'"):
        nets.AttentionModule(64, "diff")
