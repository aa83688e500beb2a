"""GPU parity of the fused policy / critic forward, the masked-aggregation autograd op and the
PPO scan kernels: against the reference's golden vectors (tolerance 1e-5, north_star) and the
CPU oracle."""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.test_oracle_golden import GOLDEN, POLICY_FIXTURES

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-5, atol=1e-5)


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need the MI355X")
    return torch


def build_nets(z, n_agents, torch):
    from com_marl_amd import nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    d_total = z["obs"].shape[1]
    spec = EnvSpec(_Box(np.zeros(d_total), np.ones(d_total)), _Discrete(5))
    hops = z["channels"].shape[1]
    res = bool(z["residual"]) if "residual" in z.files else True
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=n_agents, n_gcn_layers=hops, residual=res, device="cuda:0")
    crit = nets.CommBaseCritic(spec, n_agents=n_agents, n_gcn_layers=hops, residual=res, device="cuda:0")
    # load the REFERENCE state_dict by the reference's parameter names (strict)
    pol.load_state_dict({k[4:]: torch.as_tensor(z[k]) for k in z.files if k.startswith("pol.")}, strict=True)
    crit.load_state_dict({k[5:]: torch.as_tensor(z[k]) for k in z.files if k.startswith("crit.")}, strict=True)
    return pol, crit


@pytest.mark.parametrize("name,n_agents", POLICY_FIXTURES)
def test_fused_forward_matches_reference(name, n_agents, torch_cuda):
    torch = torch_cuda
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    pol, crit = build_nets(z, n_agents, torch)
    dev = "cuda:0"
    obs, adj, ch = (torch.as_tensor(z[k]).to(dev) for k in ("obs", "adj", "channels"))
    _, probs, attn = pol.act_device(obs, None, adj, ch)
    np.testing.assert_allclose(probs.cpu().numpy(), z["probs"], **TOL)
    np.testing.assert_allclose(attn.cpu().numpy(), z["attn"], **TOL)
    av = torch.as_tensor(z["avail_masked"]).to(dev)
    _, probs_m, _ = pol.act_device(obs, av, adj, ch)
    np.testing.assert_allclose(probs_m.cpu().numpy(), z["probs_masked"], **TOL)
    v = crit.values_device(obs, adj, ch)
    np.testing.assert_allclose(v.cpu().numpy(), z["values"], rtol=1e-5, atol=1e-5 * n_agents)
    # numpy-in / numpy-out drop-in call (reference get_actions signature), greedy
    S = obs.shape[0]
    acts, infos = pol.get_actions(z["obs"], np.ones((S, n_agents * 5), np.float32), z["adj"], z["channels"], greedy=True)
    assert acts.shape == (S, n_agents) and acts.dtype == np.int64
    np.testing.assert_array_equal(acts, z["probs"].argmax(-1))
    np.testing.assert_allclose(np.stack(infos["action_probs"]), z["probs"], **TOL)
    np.testing.assert_allclose(np.stack(infos["attention_weights"]), z["attn"], **TOL)


@pytest.mark.parametrize("name,n_agents", POLICY_FIXTURES)
def test_autograd_path_matches_reference(name, n_agents, torch_cuda):
    torch = torch_cuda
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    pol, crit = build_nets(z, n_agents, torch)
    dev = "cuda:0"
    S = z["obs"].shape[0]
    obs = torch.as_tensor(z["obs"]).to(dev)
    adj = torch.as_tensor(z["adj"].reshape(S, -1)).to(dev)                 # sampler layout [S, N*N]
    ch = torch.as_tensor(z["channels"].reshape(S, -1, n_agents)).to(dev)   # [S, L*N, N]
    av = torch.ones(S, n_agents * 5, device=dev)
    ent = pol.entropy(obs, av, adj, ch)
    ll = pol.log_likelihood(obs, av, adj, ch, torch.as_tensor(z["actions"]).to(dev))
    np.testing.assert_allclose(ent.detach().cpu().numpy(), z["entropy"], **TOL)
    np.testing.assert_allclose(ll.detach().cpu().numpy(), z["loglik"], rtol=1e-5, atol=1e-5 * n_agents)
    loss = crit.compute_loss(obs, torch.as_tensor(z["returns"]).to(dev), adj, ch)
    np.testing.assert_allclose(loss.item(), z["critic_loss"], rtol=1e-5)
    with torch.no_grad():
        v = crit.forward(obs, av, adj, ch)
    np.testing.assert_allclose(v.cpu().numpy(), z["values"], rtol=1e-5, atol=1e-5 * n_agents)


def _agg_torch(attn, adj, ch, hw, bias):
    """plain PyTorch f32 reference of the op (comm_base_net.py:101-103, graph_conv_module.py:63-70)"""
    import torch
    A = attn * adj * ch
    A = A / (A.sum(-1, keepdim=True) + 1e-12)
    return torch.tanh(torch.matmul(A, hw) + bias)


@pytest.mark.parametrize("S,N", [(37, 4), (5, 24), (3, 54), (2, 72), (9, 3), (130, 8), (4, 9), (3, 16), (2, 100), (2, 128), (2100, 24),
                                 (700, 54)])
def test_masked_aggregate_forward_backward(S, N, torch_cuda):
    torch = torch_cuda
    from com_marl_amd.nets import masked_aggregate
    g = torch.Generator(device="cpu").manual_seed(S * 100 + N)
    dev = "cuda:0"
    attn = torch.softmax(torch.randn(S, N, N, generator=g), -1).to(dev).requires_grad_()
    adj = (torch.rand(S, N, N, generator=g) < 0.7).float()
    adj[:, range(N), range(N)] = 1.0
    ch = (torch.rand(S, 2, N, N, generator=g) < 0.7).float()
    ch[:, :, range(N), range(N)] = 1.0
    adj, ch = adj.to(dev), ch.to(dev)
    hw = torch.randn(S, N, 64, generator=g).to(dev).requires_grad_()
    bias = (torch.randn(64, generator=g) * 0.1).to(dev).requires_grad_()
    w = torch.randn(S, N, 64, generator=g).to(dev)
    for hop in (0, 1):
        out = masked_aggregate(attn, adj, ch, hop, hw, bias)
        ref = _agg_torch(attn, adj, ch[:, hop], hw, bias)
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
        g_out = torch.autograd.grad((out * w).sum(), (attn, hw, bias))
        g_ref = torch.autograd.grad((ref * w).sum(), (attn, hw, bias))
        for a, b, nm in zip(g_out, g_ref, ("d_attn", "d_hw", "d_bias")):
            bn = b.cpu().numpy()          # f32 cancellation noise scales with the gradient magnitude
            np.testing.assert_allclose(a.cpu().numpy(), bn, rtol=2e-4, atol=1e-5 * max(1.0, float(np.abs(bn).max())),
                                       err_msg=f"{nm} hop{hop}")
    # None masks == all ones
    out = masked_aggregate(attn, None, None, 0, hw, bias)
    ref = _agg_torch(attn, torch.ones_like(adj), torch.ones_like(adj), hw, bias)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_sampling_matches_oracle_stream(torch_cuda):
    """Same Philox counter -> same actions as the CPU oracle's inverse-CDF sampler."""
    torch = torch_cuda
    z = np.load(os.path.join(GOLDEN, "policy_pp_map10.npz"))
    pol, _ = build_nets(z, 4, torch)
    pol.set_rng(seed=77, env_id_offset=1000)
    obs, adj, ch = (torch.as_tensor(z[k]).to("cuda:0") for k in ("obs", "adj", "channels"))
    acts, probs, _ = pol.act_device(obs, None, adj, ch, policy_step=5)
    want = O.sample_actions(probs.cpu().numpy(), 77, 1000, 5)
    np.testing.assert_array_equal(acts.cpu().numpy(), want)
    # distribution sanity on a larger batch: empirical frequencies follow probs
    big = obs.repeat(400, 1)
    counts = np.zeros((4, 5))
    for step in range(3):
        a, p, _ = pol.act_device(big, None, adj.repeat(400, 1, 1), ch.repeat(400, 1, 1, 1), policy_step=100 + step)
        a = a.cpu().numpy().reshape(400, -1, 4)[:, 0]
        for i in range(4):
            counts[i] += np.bincount(a[:, i], minlength=5)
    freq = counts / counts.sum(1, keepdims=True)
    np.testing.assert_allclose(freq, p.cpu().numpy()[0], atol=0.06)


def test_returns_gae_kernels(torch_cuda):
    torch = torch_cuda
    import ctypes as C
    from com_marl_amd import _lib as L
    z = np.load(os.path.join(GOLDEN, "ppo_math.npz"))
    dev = "cuda:0"
    lens = torch.as_tensor(z["lens"].astype(np.int32)).to(dev)
    rew64 = torch.as_tensor(z["rewards_pad"]).to(dev)
    P, T = rew64.shape
    ret = torch.empty(P, T, dtype=torch.float32, device=dev)
    L.check(L.lib().cm_discount_returns(P, T, L.ptr(rew64), L.ptr(lens), float(z["gamma"]), L.ptr(ret), None))
    np.testing.assert_array_equal(ret.cpu().numpy(), z["returns"])
    rew = rew64.float().contiguous()
    base = torch.as_tensor(z["baselines"]).to(dev)
    adv = torch.empty_like(rew)
    L.check(L.lib().cm_gae(P, T, L.ptr(rew), L.ptr(base), L.ptr(lens), float(z["gamma"]), float(z["lam"]), 0, 1e-8,
                           L.ptr(adv), None))
    np.testing.assert_allclose(adv.cpu().numpy(), z["adv"], **TOL)
    L.check(L.lib().cm_gae(P, T, L.ptr(rew), L.ptr(base), L.ptr(lens), float(z["gamma"]), float(z["lam"]), 1, 1e-8,
                           L.ptr(adv), None))
    np.testing.assert_allclose(adv.cpu().numpy(), z["adv_norm"], **TOL)
    # size-independent property at scale: oracle agreement on a ragged 3000 x 200 batch
    rng = np.random.RandomState(0)
    P, T = 3000, 200
    lens_np = rng.randint(1, T + 1, size=P).astype(np.int32)
    r = rng.randn(P, T).astype(np.float32)
    r[np.arange(T)[None, :] >= lens_np[:, None]] = 0
    b = rng.randn(P, T).astype(np.float32)
    adv = torch.empty(P, T, dtype=torch.float32, device=dev)
    r_d, b_d, l_d = torch.as_tensor(r).to(dev), torch.as_tensor(b).to(dev), torch.as_tensor(lens_np).to(dev)
    L.check(L.lib().cm_gae(P, T, L.ptr(r_d), L.ptr(b_d), L.ptr(l_d), 0.99, 0.97, 1, 1e-8, L.ptr(adv), None))
    want = O.normalize_advantages(O.gae(r, b, 0.99, 0.97), lens_np)
    # per-path normalisation divides by sqrt(var + 1e-8): very short paths amplify f32-vs-f64 rounding,
    # so this stress case uses a looser bar than the golden-vector check above (1e-5)
    np.testing.assert_allclose(adv.cpu().numpy(), want, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("S,N", [(50, 4), (7, 24), (3, 54), (2, 72), (11, 3), (130, 8), (4, 9), (3, 16), (2, 100), (2, 128), (2100, 24),
                                 (700, 54)])
def test_attention_softmax_op(S, N, torch_cuda):
    """fused scores+softmax (cm_attention_forward/backward) vs the plain PyTorch f32 ops it replaces"""
    torch = torch_cuda
    from com_marl_amd.nets import _AttentionSoftmax
    g = torch.Generator().manual_seed(S + N)
    q = (torch.randn(S, N, 64, generator=g) * 0.3).cuda().requires_grad_()
    e = (torch.randn(S, N, 64, generator=g) * 0.3).cuda().requires_grad_()
    w = torch.randn(S, N, N, generator=g).cuda()
    m = _AttentionSoftmax.apply(q, e)
    ref = torch.softmax(torch.matmul(q, e.transpose(-2, -1)), dim=-1)
    np.testing.assert_allclose(m.detach().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    g1 = torch.autograd.grad((m * w).sum(), (q, e))
    g2 = torch.autograd.grad((ref * w).sum(), (q, e))
    for a, b in zip(g1, g2):
        bn = b.cpu().numpy()
        np.testing.assert_allclose(a.cpu().numpy(), bn, rtol=2e-4, atol=1e-5 * max(1.0, float(np.abs(bn).max())))


@pytest.mark.parametrize("R,IN,OUT", [(1000, 21, 128), (70001, 128, 64), (4097, 64, 64), (333, 64, 128), (5000, 32, 5),
                                      (129, 64, 1), (64, 77, 128), (100000, 128, 128)])
def test_linear_weight_gradient_kernel(R, IN, OUT, torch_cuda):
    """cm_linear_wgrad (f32 MFMA, atomics merge) vs torch autograd for nn.Linear and the GCN H.W product"""
    torch = torch_cuda
    from com_marl_amd.nets import HipLinear, _MatmulWFn
    g = torch.Generator().manual_seed(R + IN + OUT)
    x = torch.randn(R, IN, generator=g).cuda()
    dy = torch.randn(R, OUT, generator=g).cuda()
    lin = HipLinear(IN, OUT).cuda()
    xr = x.clone().requires_grad_()
    y = lin(xr)
    y.backward(dy)
    ref = torch.nn.Linear(IN, OUT).cuda()
    ref.load_state_dict(lin.state_dict())
    xr2 = x.clone().requires_grad_()
    ref(xr2).backward(dy)
    scale = float(ref.weight.grad.abs().max())
    np.testing.assert_allclose(lin.weight.grad.cpu().numpy(), ref.weight.grad.cpu().numpy(), rtol=2e-4, atol=2e-5 * scale)
    np.testing.assert_allclose(lin.bias.grad.cpu().numpy(), ref.bias.grad.cpu().numpy(), rtol=2e-4,
                               atol=2e-5 * float(ref.bias.grad.abs().max()))
    np.testing.assert_allclose(xr.grad.cpu().numpy(), xr2.grad.cpu().numpy(), rtol=1e-5, atol=1e-5)
    if IN == OUT == 64:
        w = torch.randn(64, 64, generator=g).cuda().requires_grad_()
        h = x.reshape(-1, 1, 64).clone().requires_grad_()
        _MatmulWFn.apply(h, w).backward(dy.reshape(-1, 1, 64))
        w2 = w.detach().clone().requires_grad_()
        torch.matmul(x, w2).backward(dy)
        np.testing.assert_allclose(w.grad.cpu().numpy(), w2.grad.cpu().numpy(), rtol=2e-4, atol=2e-5 * float(w2.grad.abs().max()))


@pytest.mark.parametrize("R,IN,OUT,act,layout", [(1000, 21, 128, 1, 0), (70001, 128, 64, 1, 0), (4097, 64, 64, 0, 1),
                                                  (333, 64, 128, 1, 0), (5000, 32, 5, 0, 0), (129, 64, 1, 0, 0),
                                                  (64, 77, 128, 1, 0), (100000, 128, 128, 1, 0), (63, 64, 64, 0, 0),
                                                  (300001, 64, 32, 1, 0), (20000, 32, 64, 1, 0), (20001, 32, 32, 0, 0), (12345, 32, 128, 1, 0),
                                                  (5000, 128, 32, 0, 0), (40961, 64, 64, 1, 1), (16447, 64, 128, 0, 0), (30001, 53, 128, 1, 0), (5000, 29, 64, 0, 0),
                                                  (70003, 77, 128, 1, 0), (129, 21, 32, 1, 0)])
def test_fused_linear_act_kernels(R, IN, OUT, act, layout, torch_cuda):
    """cm_linear_act_forward / backward (one HBM pass each: y = act(x W^T + b); dx, dW, db from dy, y, x) against an f64
    torch reference of the same layer - forward and every gradient, both weight layouts, ragged last chunks.
    Tolerances: forward 1e-5; gradients relative to the largest entry of the f64 result (sums over up to 3e5 rows of
    f32 products, merged across workgroups with float atomics: order-dependent in the last bits)."""
    torch = torch_cuda
    from com_marl_amd.nets import _LinearActFn
    g = torch.Generator().manual_seed(R + IN + OUT)
    x = torch.randn(R, IN, generator=g).cuda()
    dy = torch.randn(R, OUT, generator=g).cuda()
    w = (torch.randn(OUT, IN, generator=g) * 0.2).cuda() if layout == 0 else (torch.randn(IN, OUT, generator=g) * 0.2).cuda()
    b = (torch.randn(OUT, generator=g) * 0.1).cuda() if layout == 0 else None
    xr, wr = x.clone().requires_grad_(), w.clone().requires_grad_()
    br = None if b is None else b.clone().requires_grad_()
    y = _LinearActFn.apply(xr, wr, br, act, layout)
    y.backward(dy)
    x6, w6 = x.double().requires_grad_(), w.double().requires_grad_()
    b6 = None if b is None else b.double().requires_grad_()
    z = x6 @ (w6.t() if layout == 0 else w6)
    if b6 is not None:
        z = z + b6
    y6 = torch.tanh(z) if act else z
    y6.backward(dy.double())
    np.testing.assert_allclose(y.detach().cpu().numpy(), y6.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(xr.grad.cpu().numpy(), x6.grad.cpu().numpy(), rtol=1e-5, atol=2e-5)
    for got, want in ((wr.grad, w6.grad), (None if br is None else br.grad, None if b6 is None else b6.grad)):
        if want is None:
            continue
        scale = float(want.abs().max())
        np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=2e-6 * scale)
    # first layer form: no input gradient requested
    wr2 = w.clone().requires_grad_()
    _LinearActFn.apply(x, wr2, None if b is None else b, act, layout).backward(dy)
    np.testing.assert_allclose(wr2.grad.cpu().numpy(), w6.grad.cpu().numpy(), rtol=1e-5, atol=2e-6 * float(w6.grad.abs().max()))
    # a second gradient into the same output, summed inside the kernel (dy2) == the caller adding first
    from com_marl_amd.nets import _lin_bwd
    dy_b = torch.randn(R, OUT, generator=g).cuda()
    yv = y.detach() if act else None
    dx2, dw2, _ = _lin_bwd(x, w, layout, dy, yv, True, b is not None, dy_add=dy_b)
    dx1, dw1, _ = _lin_bwd(x, w, layout, dy + dy_b, yv, True, b is not None)
    np.testing.assert_allclose(dx2.cpu().numpy(), dx1.cpu().numpy(), rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(dw2.cpu().numpy(), dw1.cpu().numpy(), rtol=1e-5, atol=2e-6 * float(dw1.abs().max()))


@pytest.mark.parametrize("d,n_agents,residual,hops", [(21, 4, False, 2), (100, 4, True, 2), (100, 6, False, 1), (29, 3, True, 3),
                                                      (77, 24, False, 2), (53, 72, True, 1), (21, 8, True, 2), (29, 16, True, 2),
                                                      (53, 20, False, 2), (77, 32, True, 2), (12, 36, True, 2), (40, 80, False, 3),
                                                      (21, 4, True, 0), (77, 54, True, 0), (100, 5, True, 0), (21, 4, False, 0)])
def test_fused_kernels_vs_autograd_path(d, n_agents, residual, hops, torch_cuda):
    """residual on/off, hop counts, odd team sizes and obs dims without an MFMA build (d=100 -> the generic
    VALU kernel): the fused forward must equal the autograd path (torch GEMMs + masked_aggregate)."""
    torch = torch_cuda
    from com_marl_amd import nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    torch.manual_seed(d + n_agents)
    spec = EnvSpec(_Box(np.zeros(d * n_agents), np.ones(d * n_agents)), _Discrete(5))
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=n_agents, residual=residual, n_gcn_layers=hops, device="cuda:0")
    crit = nets.CommBaseCritic(spec, n_agents=n_agents, residual=residual, n_gcn_layers=hops, device="cuda:0")
    S = 37
    obs = torch.rand(S, n_agents * d, device="cuda:0")
    adj = (torch.rand(S, n_agents, n_agents, device="cuda:0") < 0.6).float()
    adj[:, range(n_agents), range(n_agents)] = 1
    ch = (torch.rand(S, hops, n_agents, n_agents, device="cuda:0") < 0.7).float()
    ch[:, :, range(n_agents), range(n_agents)] = 1
    _, probs, attn = pol.act_device(obs, None, adj, ch)
    with torch.no_grad():
        p_ref, a_ref = pol._probs(obs, None, adj, ch)
        v_ref, _ = crit._values_grad(obs, adj, ch)
    np.testing.assert_allclose(probs.cpu().numpy(), p_ref.cpu().numpy(), **TOL)
    np.testing.assert_allclose(attn.cpu().numpy(), a_ref.cpu().numpy(), **TOL)
    v = crit.values_device(obs, adj, ch)
    np.testing.assert_allclose(v.cpu().numpy(), v_ref.cpu().numpy(), rtol=1e-5, atol=1e-5 * n_agents)


@pytest.mark.parametrize("n_agents,d", [(4, 21), (24, 77), (54, 77)])
def test_operand_pack_path_equals_plain_weight_path(n_agents, d, torch_cuda):
    """cm_policy_forward with the matrix-core operand pack (MFMA kernels) vs without it (mfma_pack = NULL: the generic
    VALU kernel reads the plain [in,out] weights): same net, same inputs, results within the f32 reordering error."""
    import ctypes as C
    torch = torch_cuda
    from com_marl_amd import _lib as L, nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    torch.manual_seed(n_agents)
    spec = EnvSpec(_Box(np.zeros(d * n_agents), np.ones(d * n_agents)), _Discrete(5))
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=n_agents, device="cuda:0")
    S = 19
    obs = torch.rand(S, n_agents * d, device="cuda:0")
    _, p_pack, a_pack = pol.act_device(obs, None, None, None, want_actions=False)
    w = pol._weights_struct()
    assert w.mfma_pack and L.lib().cm_policy_pack_bytes(C.byref(w)) == pol._mfma.numel() * 4
    w.mfma_pack = None
    probs = torch.empty(S, n_agents, 5, device="cuda:0")
    attn = torch.empty(S, n_agents, n_agents, device="cuda:0")
    L.check(L.lib().cm_policy_forward(C.byref(w), S, obs.data_ptr(), None, None, None, 1, 0, 0, None, 0, None,
                                      probs.data_ptr(), attn.data_ptr(), None), "cm_policy_forward (plain weights)")
    torch.cuda.synchronize()
    np.testing.assert_allclose(p_pack.cpu().numpy(), probs.cpu().numpy(), **TOL)
    np.testing.assert_allclose(a_pack.cpu().numpy(), attn.cpu().numpy(), **TOL)
    # the pack follows in-place weight updates (same buffer, rewritten by sync_weights)
    ptr0 = pol._mfma.data_ptr()
    with torch.no_grad():
        for prm in pol.parameters():
            prm.mul_(0.5)
    _, p2, _ = pol.act_device(obs, None, None, None, want_actions=False)
    assert pol._mfma.data_ptr() == ptr0 and (p2 - p_pack).abs().max() > 1e-4
    with torch.no_grad():
        p_ref, _ = pol._probs(obs, None, None, None)
    np.testing.assert_allclose(p2.cpu().numpy(), p_ref.cpu().numpy(), **TOL)


@pytest.mark.parametrize("d,hops,residual,masks,N", [(21, 2, True, False, 4), (21, 2, True, True, 4), (53, 1, False, True, 4), (77, 3, True, True, 4),
                                                    (29, 2, False, False, 4), (21, 2, True, True, 6), (29, 2, False, True, 3),
                                                    (77, 2, True, True, 24), (53, 2, True, True, 72), (77, 1, False, False, 54),
                                                    (21, 2, True, True, 16), (40, 3, True, True, 80), (29, 2, True, True, 12), (21, 1, True, True, 9),
                                                    (21, 2, True, False, 96)])
def test_fused_training_path_vs_per_layer_autograd(d, hops, residual, masks, N, torch_cuda, monkeypatch):
    """Every team size: the whole-network training path (ONE forward launch that stores the activations, cm_*_forward_saved, +
    the hand-written backward chain of nets._FusedNetFn) against the per-layer autograd path of the same nets
    (COMMARL_FUSED_TRAIN=0): policy probabilities / critic values 1e-5, every parameter gradient to 1e-4 relative + 1e-5
    of the tensor's largest entry (the two forwards differ in the last bits: f16-split MFMA vs f32 MFMA per layer)."""
    torch = torch_cuda
    from com_marl_amd import nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    P, T = (7, 13) if N <= 24 else (3, 5)                  # (N = 96: above the fused path's limit -> both modes run per layer)
    spec = EnvSpec(_Box(np.zeros(N * d), np.ones(N * d)), _Discrete(5))
    torch.manual_seed(d + hops)
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, n_gcn_layers=hops, residual=residual, device="cuda:0")
    crit = nets.CommBaseCritic(spec, n_agents=N, n_gcn_layers=hops, residual=residual, device="cuda:0")
    for net in (pol, crit):
        for n_, p_ in net.named_parameters():
            if n_.endswith("bias"):
                torch.nn.init.uniform_(p_, -0.2, 0.2)     # non-zero biases so their handling is pinned
    g = torch.Generator().manual_seed(5)
    obs = torch.rand(P, T, N * d, generator=g).cuda()
    adj = ch = None
    if masks:
        adj = (torch.rand(P, T, N, N, generator=g) < 0.7).float().cuda()
        adj[..., torch.arange(N), torch.arange(N)] = 1.0
        ch = (torch.rand(P, T, hops, N, N, generator=g) < 0.8).float().cuda()
    avail = (torch.rand(P, T, N * 5, generator=g) < 0.85).float().cuda()
    avail.view(P, T, N, 5)[..., 0] = 1.0
    actions = torch.randint(0, 5, (P, T, N), generator=g).cuda()
    wts = torch.randn(P, T, generator=g).cuda()
    returns = torch.randn(P, T, generator=g).cuda()
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("COMMARL_FUSED_TRAIN", mode)
        pol.zero_grad(); crit.zero_grad()
        probs, attn = pol._probs(obs, avail, adj, ch)
        dist = torch.distributions.Categorical(probs=probs)
        loss = ((dist.log_prob(actions).sum(-1) + 0.1 * dist.entropy().mean(-1)) * wts).sum()
        loss.backward()
        closs = crit.compute_loss(obs, returns, adj, ch)
        closs.backward()
        res[mode] = dict(probs=probs.detach().cpu().numpy(), attn=attn.detach().cpu().numpy(), closs=float(closs.detach()),
                         gp={n_: p_.grad.cpu().numpy().copy() for n_, p_ in pol.named_parameters()},
                         gc={n_: p_.grad.cpu().numpy().copy() for n_, p_ in crit.named_parameters() if p_.grad is not None})
    a, b = res["1"], res["0"]
    np.testing.assert_allclose(a["probs"], b["probs"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a["attn"], b["attn"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a["closs"], b["closs"], rtol=1e-5)
    assert set(a["gc"]) == set(b["gc"]) and set(a["gp"]) == set(b["gp"])
    for key in ("gp", "gc"):
        for n_ in b[key]:
            scale = max(1e-6, float(np.abs(b[key][n_]).max()))
            np.testing.assert_allclose(a[key][n_], b[key][n_], rtol=1e-4, atol=1e-5 * scale, err_msg=f"{key} {n_}")


@pytest.mark.parametrize("hops,residual,masks,P,T", [(2, True, False, 7, 13), (2, True, True, 5, 64), (1, False, True, 3, 11)])
def test_wave_owned_training_forward_matches_the_workgroup_tiled_one(hops, residual, masks, P, T, torch_cuda, monkeypatch):
    """Teams of 4, large batches: the training forward on the rollout's wave-owned kernel (cm_policy_forward_saved_wave: a
    persistent workgroup per CU, saves written from the epilogue registers) against cm_policy_forward_saved - same logits and
    attention to 1e-5, every parameter gradient of the same backward chain to 1e-4 relative + 1e-5 of the tensor's largest entry
    (the two forwards sum in another order).  Env counts that leave the last workgroup ragged and that wrap the persistent loop."""
    torch = torch_cuda
    from com_marl_amd import nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    N, d = 4, 21
    spec = EnvSpec(_Box(np.zeros(N * d), np.ones(N * d)), _Discrete(5))
    torch.manual_seed(hops)
    pol = nets.CommCategoricalMLPPolicy(spec, n_agents=N, n_gcn_layers=hops, residual=residual, device="cuda:0")
    crit = nets.CommBaseCritic(spec, n_agents=N, n_gcn_layers=hops, residual=residual, device="cuda:0")
    for net in (pol, crit):
        for n_, p_ in net.named_parameters():
            if n_.endswith("bias"):
                torch.nn.init.uniform_(p_, -0.2, 0.2)
    g = torch.Generator().manual_seed(9)
    obs = torch.rand(P, T, N * d, generator=g).cuda()
    adj = ch = None
    if masks:
        adj = (torch.rand(P, T, N, N, generator=g) < 0.7).float().cuda()
        adj[..., torch.arange(N), torch.arange(N)] = 1.0
        ch = (torch.rand(P, T, hops, N, N, generator=g) < 0.8).float().cuda()
    actions = torch.randint(0, 5, (P, T, N), generator=g).cuda()
    wts = torch.randn(P, T, generator=g).cuda()
    returns = torch.randn(P, T, generator=g).cuda()
    res = {}
    for tag, min_envs in (("wave", "1"), ("tiled", "1000000000")):
        monkeypatch.setenv("COMMARL_TRAIN_FWD_WAVE_MIN", min_envs)
        pol.zero_grad(); crit.zero_grad()
        logits = pol._logits(obs, adj, ch)
        dist = torch.distributions.Categorical(logits=logits)
        loss = ((dist.log_prob(actions).sum(-1) + 0.1 * dist.entropy().mean(-1)) * wts).sum()
        loss.backward()
        values, _ = crit._values_grad(obs, adj, ch)
        closs = crit.compute_loss(obs, returns, adj, ch)
        closs.backward()
        res[tag] = dict(logits=logits.detach().cpu().numpy(), values=values.detach().cpu().numpy(), closs=float(closs.detach()),
                        gp={n_: p_.grad.cpu().numpy().copy() for n_, p_ in pol.named_parameters()},
                        gc={n_: p_.grad.cpu().numpy().copy() for n_, p_ in crit.named_parameters() if p_.grad is not None})
    a, b = res["wave"], res["tiled"]
    np.testing.assert_allclose(a["logits"], b["logits"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a["values"], b["values"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(a["closs"], b["closs"], rtol=1e-5)
    assert np.abs(a["logits"] - b["logits"]).max() > 0 and np.abs(a["values"] - b["values"]).max() > 0, "both runs took the same kernel"
    for key in ("gp", "gc"):
        for n_ in b[key]:
            scale = max(1e-6, float(np.abs(b[key][n_]).max()))
            np.testing.assert_allclose(a[key][n_], b[key][n_], rtol=1e-4, atol=1e-5 * scale, err_msg=f"{key} {n_}")


def test_evaluate_nograd_shares_one_forward(torch_cuda):
    """policy.evaluate_nograd (ONE launch giving logits and action probabilities: what train_once shares between the
    loss, the old log-likelihood and the KL / entropy diagnostics) returns the probabilities of act_device and logits whose
    softmax they are.  Bit for bit where act_device runs the same workgroup-tiled body (every team size but 4); teams of 4 act
    through the wave-owned kernel (cm_policy_w_dev.h: same arithmetic scheme, another summation order), f32-grade agreement.
    (train_once takes BOTH sides of its probability ratio from evaluate_nograd / the training forward, never from act_device.)"""
    torch = torch_cuda
    from com_marl_amd import nets
    from com_marl_amd.envs import EnvSpec, _Box, _Discrete
    torch.manual_seed(5)
    for n_agents, d in ((4, 21), (6, 21)):
        spec = EnvSpec(_Box(np.zeros(d * n_agents), np.ones(d * n_agents)), _Discrete(5))
        pol = nets.CommCategoricalMLPPolicy(spec, n_agents=n_agents, device="cuda:0")
        P, T = 7, 13
        obs = torch.rand(P, T, n_agents * d, device="cuda:0")
        adj = (torch.rand(P, T, n_agents, n_agents, device="cuda:0") < 0.6).float()
        ch = (torch.rand(P, T, 2, n_agents, n_agents, device="cuda:0") < 0.7).float()
        logits, probs = pol.evaluate_nograd(obs, adj, ch)
        _, ref, _ = pol.act_device(obs.reshape(P * T, -1), None, adj.reshape(P * T, n_agents, n_agents),
                                   ch.reshape(P * T, 2, n_agents, n_agents), want_actions=False, want_attn=False, policy_step=0)
        assert probs.shape == (P, T, n_agents, 5)
        if n_agents == 4 and os.environ.get("COMMARL_POLICY_KERNEL", "w")[:1] not in ("h", "f", "v"):
            np.testing.assert_allclose(probs.reshape(P * T, n_agents, 5).cpu().numpy(), ref.cpu().numpy(), rtol=3e-6, atol=3e-7)
        else:
            assert torch.equal(probs.reshape(P * T, n_agents, 5), ref)
        assert logits is not None and logits.shape == (P, T, n_agents, 5)
        np.testing.assert_allclose(torch.softmax(logits, -1).cpu().numpy(), probs.cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_weight_pack_refuses_weights_outside_the_f16_range():
    """cm_policy_pack / cm_critic_pack: a weight the (hi, lo) f16 pair cannot carry (|w| > 65504 - 22 700 where the wave-owned
    pack folds the tanh prescale in -, inf, NaN) is refused with a
    status < 0 and a message, not packed as +-inf; the same net packs again once the weight is back in range."""
    import torch
    from com_marl_amd import _lib as L, envs as E, nets
    spec = E.EnvSpec(E._Box(np.zeros(84), np.ones(84)), E._Discrete(5))
    torch.manual_seed(0)
    for net in (nets.CommCategoricalMLPPolicy(spec, n_agents=4, device="cuda:0"), nets.CommBaseCritic(spec, n_agents=4, device="cuda:0")):
        net.sync_weights()                                            # fine as initialised
        w = net.gcn_layers[1].weight
        keep = w.detach().clone()
        for bad in (7.0e4, -1.0e5, float("inf"), float("nan")):
            with torch.no_grad():
                w[3, 5] = bad
            with pytest.raises(L.CommarlError, match="outside the f16 range"):
                net.sync_weights()
        with torch.no_grad():
            w.copy_(keep)
            w[3, 5] = 2.0e4                                           # a large weight inside the range is fine (tanh layers carry
                                                                      # 2 log2(e) w in the wave-owned pack: the limit there is 65504 / 2.885)
        net.sync_weights()
        with torch.no_grad():
            w.copy_(keep)
        net.sync_weights()
