"""Comm-DP policy and critic: same classes, ctor kwargs, parameter names and numerics as the
reference, two execution paths:

  * rollout / no-grad : ONE fused HIP launch per batch of env states through the C ABI
    (``cm_policy_forward`` / ``cm_critic_forward``, csrc/cm_policy.hip);
  * PPO update (autograd): dense per-agent GEMMs on PyTorch-ROCm (rocBLAS / MFMA), the
    adjacency-masked aggregation as the custom HIP op ``masked_aggregate``
    (``cm_masked_agg_forward/backward``, csrc/cm_ppo.hip).

Reference: com_marl/torch/modules/comm_base_net.py, attention_module.py, graph_conv_module.py,
mlp_encoder_module.py, categorical_mlp_module.py, gaussian_mlp_module.py,
policies/comm_categorical_mlp_policy.py, baselines/comm_base_critic.py,
garage/torch/modules/multi_headed_mlp_module.py.
"""
import ctypes as C
import math
import os
from collections import OrderedDict

import numpy as np
import torch
from torch import nn
from torch.distributions import Categorical, Normal

from . import _lib as L


MAX_KERNEL_AGENTS = 128      # team size up to which the N x N HIP kernels hold one env's matrix in LDS
MAX_FUSED_AGENTS = 80        # ... and up to which the one-launch forward (rollout and training) does (cm_policy_h_dev.h LDS map)

# ---------------------------------------------------------------------------------------------
# building blocks with the reference's state_dict names
# ---------------------------------------------------------------------------------------------
class _Tanh(nn.Module):
    def forward(self, x):
        return torch.tanh(x)


class MLPModule(nn.Module):
    """garage MultiHeadedMLPModule with one head (multi_headed_mlp_module.py:54-149):
    ``_layers.{i}.linear`` (+tanh) then ``_output_layers.0.linear`` (+ optional nonlinearity).
    Init order mirrors the reference: nn.Linear default init, then xavier_uniform_ / zeros_."""

    def __init__(self, input_dim, output_dim, hidden_sizes, output_tanh=False):
        super().__init__()
        self._layers = nn.ModuleList()
        prev = input_dim
        for size in hidden_sizes:
            lin = HipLinear(prev, size)
            nn.init.xavier_uniform_(lin.weight)
            nn.init.zeros_(lin.bias)
            self._layers.append(nn.Sequential(OrderedDict(linear=lin, non_linearity=_Tanh())))
            prev = size
        lin = HipLinear(prev, output_dim)
        nn.init.xavier_uniform_(lin.weight)
        nn.init.zeros_(lin.bias)
        mods = OrderedDict(linear=lin)
        if output_tanh:
            mods["non_linearity"] = _Tanh()
        self._output_layers = nn.ModuleList([nn.Sequential(mods)])

    def forward(self, x):
        for layer in self._layers:
            x = hip_linear(x, layer.linear, act=1)                       # Sequential(linear, tanh) as one fused op
        out = self._output_layers[0]
        return hip_linear(x, out.linear, act=1 if len(out) > 1 else 0)


class _AttentionSoftmax(torch.autograd.Function):
    """M = softmax_j(q_i . e_j) per sample, one HIP kernel each way instead of three batched N x N GEMMs."""

    @staticmethod
    def forward(ctx, q, e):
        S, N, E = e.shape
        q, e = q.contiguous(), e.contiguous()
        m = torch.empty(S, N, N, dtype=e.dtype, device=e.device)
        with torch.cuda.device(e.device):
            L.check(L.lib().cm_attention_forward(S, N, E, L.ptr(q), L.ptr(e), L.ptr(m), L.current_stream()),
                    "cm_attention_forward")
        ctx.save_for_backward(q, e, m)
        return m

    @staticmethod
    def backward(ctx, d_m):
        q, e, m = ctx.saved_tensors
        S, N, E = e.shape
        d_m = d_m.contiguous()
        d_q, d_e = torch.empty_like(q), torch.empty_like(e)
        with torch.cuda.device(e.device):
            L.check(L.lib().cm_attention_backward(S, N, E, L.ptr(q), L.ptr(e), L.ptr(m), L.ptr(d_m), None, None, L.ptr(d_q),
                                                  L.ptr(d_e), L.current_stream()), "cm_attention_backward")
        return d_q, d_e


class AttentionModule(nn.Module):
    """attention_module.py:17-51: 'general' softmax_j((q W^T) . k_j) with the learned ``linear_in``, or 'dot'
    softmax_j(q . k_j) with no parameter at all (:38-41).  ('diff' creates a parameter in the reference's constructor but its
    forward has no branch for it - it cannot run there either.)"""

    def __init__(self, dimensions, attention_type="general"):
        super().__init__()
        if attention_type not in ("general", "dot"):
            raise NotImplementedError("attention_type must be 'general' or 'dot' ('diff' has no forward branch in the reference, "
                                      "attention_module.py:36-49)")
        self.attention_type = attention_type
        if attention_type == "general":
            self.linear_in = HipLinear(dimensions, dimensions, bias=False)
        self._dim = dimensions

    def forward(self, query):
        q = self.linear_in(query) if self.attention_type == "general" else query
        if query.is_cuda and query.dim() == 3 and query.shape[-1] == 64 and query.shape[-2] <= MAX_KERNEL_AGENTS:
            return _AttentionSoftmax.apply(q, query)             # fused HIP op (cm_attention_forward/backward)
        return torch.softmax(torch.matmul(q, query.transpose(-2, -1)), dim=-1)   # teams above 128 agents (maps >= 50): library GEMMs


class GraphConvolutionModule(nn.Module):
    """graph_conv_module.py:24-72: weight [in,out] ~ U(+-1/sqrt(out)), bias likewise."""

    def __init__(self, in_features, out_features, bias=True, id=None):
        super().__init__()
        self.in_features, self.out_features, self.id = in_features, out_features, id
        self.weight = nn.Parameter(torch.empty(in_features, out_features))
        self.bias = nn.Parameter(torch.empty(out_features)) if bias else None
        stdv = 1.0 / math.sqrt(out_features)
        with torch.no_grad():
            self.weight.uniform_(-stdv, stdv)
            if self.bias is not None:
                self.bias.uniform_(-stdv, stdv)


def _wgrad(a2d, b2d, want_colsum):
    """c[p][q] = sum_r a[r][p] b[r][q] (+ column sums of a) on the MFMA weight-gradient kernel."""
    R, P = a2d.shape
    Q = b2d.shape[1]
    c = torch.zeros(P, Q, dtype=torch.float32, device=a2d.device)
    cs = torch.zeros(P, dtype=torch.float32, device=a2d.device) if want_colsum else None
    with torch.cuda.device(a2d.device):
        L.check(L.lib().cm_linear_wgrad(R, P, Q, L.ptr(a2d), L.ptr(b2d), L.ptr(c), L.ptr(cs), L.current_stream()),
                "cm_linear_wgrad")
    return c, cs


class _LinearFn(torch.autograd.Function):
    """y = x W^T + b with the weight / bias gradients on cm_linear_wgrad: over ~1e6 agent rows the library
    GEMM for dW = dY^T X (tiny output, million-deep reduction) ran at ~10 TFLOP/s and dominated the update."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return torch.nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = dy.matmul(weight) if ctx.needs_input_grad[0] else None
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        dw, db = _wgrad(dy2, x2, ctx.has_bias)
        return dx, dw, db


class _MatmulWFn(torch.autograd.Function):
    """z = h W (GraphConvolutionModule, weight [in,out]) with dW on cm_linear_wgrad."""

    @staticmethod
    def forward(ctx, h, weight):
        ctx.save_for_backward(h, weight)
        return torch.matmul(h, weight)

    @staticmethod
    def backward(ctx, dz):
        h, weight = ctx.saved_tensors
        dh = dz.matmul(weight.t()) if ctx.needs_input_grad[0] else None
        dw, _ = _wgrad(h.reshape(-1, h.shape[-1]).contiguous(), dz.reshape(-1, dz.shape[-1]).contiguous(), False)
        return dh, dw


class _LinearActFn(torch.autograd.Function):
    """y = act(x W^T + b) (layout 0, nn.Linear) or act(x W) (layout 1, GraphConvolution weight) as ONE kernel each way
    (csrc/cm_linear.hip): forward reads x and writes y; backward reads dy, y, x and writes dx while the weight and
    bias gradients accumulate in registers - instead of torch's GEMM, tanh, tanh', input-gradient GEMM and
    weight-gradient GEMM, each a full HBM pass over the [rows, 64..128] activations of ~1e6 agent rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, layout):
        K = x.shape[-1]
        O = weight.shape[0] if layout == 0 else weight.shape[1]
        x2 = x.reshape(-1, K).contiguous()
        w = weight.contiguous()
        y = torch.empty(x2.shape[0], O, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            L.check(L.lib().cm_linear_act_forward(x2.shape[0], K, O, L.ptr(x2), L.ptr(w), layout,
                                                  L.ptr(None if bias is None else bias.contiguous()), int(act), L.ptr(y),
                                                  L.current_stream()), "cm_linear_act_forward")
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.meta = (tuple(x.shape), K, O, int(act), layout, bias is not None)
        return y.reshape(*x.shape[:-1], O)

    @staticmethod
    def backward(ctx, dy):
        x2, w, y = ctx.saved_tensors
        shape, K, O, act, layout, has_bias = ctx.meta
        dy2 = dy.reshape(-1, O).contiguous()
        dx = torch.empty_like(x2) if ctx.needs_input_grad[0] else None
        dw = torch.zeros_like(w)
        db = torch.zeros(O, dtype=torch.float32, device=w.device) if has_bias else None
        with torch.cuda.device(w.device):
            L.check(L.lib().cm_linear_act_backward(x2.shape[0], K, O, L.ptr(x2), L.ptr(w), layout, L.ptr(dy2), None, L.ptr(y),
                                                   L.ptr(dx), L.ptr(dw), L.ptr(db), L.current_stream()),
                    "cm_linear_act_backward")
        return (None if dx is None else dx.reshape(shape)), dw, db, None, None


def _fused_ok(x, weight):
    return (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and weight.requires_grad
            and max(weight.shape) <= 128 and os.environ.get("COMMARL_FUSED_LINEAR", "1") != "0")


def hip_linear(x, lin, act=0):
    """nn.Linear (+ tanh when act) forward; on the GPU with autograd on it is the fused one-pass kernel pair."""
    if _fused_ok(x, lin.weight):
        return _LinearActFn.apply(x, lin.weight, lin.bias, act, 0)
    if x.is_cuda and torch.is_grad_enabled() and lin.weight.requires_grad and max(lin.weight.shape) <= 128:
        y = _LinearFn.apply(x, lin.weight, lin.bias)
    else:
        y = torch.nn.functional.linear(x, lin.weight, lin.bias)
    return torch.tanh(y) if act else y


class HipLinear(nn.Linear):
    """nn.Linear (same parameters / state_dict) whose training backward runs on cm_linear_wgrad."""

    def forward(self, x):
        return hip_linear(x, self)


class _MaskedAggregate(torch.autograd.Function):
    """out = tanh(A.(HW) + b), A = M*R*C row-renormalised (comm_base_net.py:101-103,
    graph_conv_module.py:63-70) as one HIP kernel each way."""

    @staticmethod
    def forward(ctx, attn, dist_adj, chan_all, hop, hw, bias):
        S, N, E = hw.shape
        attn, hw = attn.contiguous(), hw.contiguous()
        out = torch.empty_like(hw)
        chan_ptr, stride = None, 0
        if chan_all is not None:
            Lh = chan_all.shape[1]
            chan_ptr, stride = chan_all.data_ptr() + 4 * hop * N * N, Lh * N * N
        with torch.cuda.device(hw.device):
            L.check(L.lib().cm_masked_agg_forward(S, N, E, L.ptr(attn), L.ptr(dist_adj), chan_ptr, stride, L.ptr(hw),
                                                  L.ptr(bias), L.ptr(out), L.current_stream()), "cm_masked_agg_forward")
        ctx.save_for_backward(attn, dist_adj, chan_all, hw, out)
        ctx.hop, ctx.has_bias = hop, bias is not None
        return out

    @staticmethod
    def backward(ctx, d_out):
        attn, dist_adj, chan_all, hw, out = ctx.saved_tensors
        S, N, E = hw.shape
        d_out = d_out.contiguous()
        d_attn, d_hw = torch.empty_like(attn), torch.empty_like(hw)
        d_bias = torch.zeros(E, dtype=hw.dtype, device=hw.device) if ctx.has_bias else None
        chan_ptr, stride = None, 0
        if chan_all is not None:
            chan_ptr, stride = chan_all.data_ptr() + 4 * ctx.hop * N * N, chan_all.shape[1] * N * N
        with torch.cuda.device(hw.device):
            L.check(L.lib().cm_masked_agg_backward(S, N, E, L.ptr(attn), L.ptr(dist_adj), chan_ptr, stride, L.ptr(hw),
                                                   L.ptr(out), None, L.ptr(d_out), L.ptr(d_attn), L.ptr(d_hw),
                                                   L.ptr(d_bias), L.current_stream()), "cm_masked_agg_backward")
        return d_attn, None, None, None, d_hw, d_bias


def masked_aggregate(attn, dist_adj, channels, hop, hw, bias):
    """attn [S,N,N], dist_adj [S,N,N] or None (= ones), channels [S,L,N,N] or None, hw [S,N,E]."""
    if not hw.is_cuda:
        raise L.CommarlError("masked_aggregate is a HIP op: tensors must live on the MI355X (no CPU fallback)")
    if hw.shape[1] > MAX_KERNEL_AGENTS:
        # teams above 128 agents (PP map 50: N = 200): the N x N tile of one env no longer fits a workgroup's LDS - the same
        # arithmetic (comm_base_net.py:101-103, graph_conv_module.py:63-70) on the framework's batched GEMM, still on the GPU
        A = attn
        if dist_adj is not None:
            A = A * dist_adj
        if channels is not None:
            A = A * channels[:, hop]
        A = A / (A.sum(dim=-1, keepdim=True) + 1e-12)
        out = torch.matmul(A, hw)
        return torch.tanh(out + bias if bias is not None else out)
    return _MaskedAggregate.apply(attn, dist_adj, channels, hop, hw, bias)


# ---------------------------------------------------------------------------------------------
# whole-network training forward (every team size <= 128): ONE fused launch that stores what the backward needs, and a hand-written
# backward chain over the C-ABI kernels - no per-layer forward kernels, no gradient-accumulation adds from autograd
# ---------------------------------------------------------------------------------------------
class _ZeroPool:
    """The weight / bias gradients of one backward pass as views of ONE zero-filled buffer (one fill launch instead of one per
    tensor; the kernels accumulate into them with float atomics, so they must start at zero).  Views are 16-byte aligned."""

    def __init__(self, params, device, extra=0):
        total = sum((p.numel() + 3) & ~3 for p in params) + extra
        self.buf = torch.zeros(total, dtype=torch.float32, device=device)
        self.off = 0

    def take(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        if self.off + n > self.buf.numel():                        # (not reached: sized from the net's parameter list)
            return torch.zeros(*shape, dtype=torch.float32, device=self.buf.device)
        v = self.buf[self.off:self.off + n].view(*shape)
        self.off += (n + 3) & ~3
        return v


def _lin_bwd(x2, w, layout, dy2, y2, want_dx, has_bias, dy_add=None, pool=None):
    """cm_linear_act_backward on 2-D contiguous tensors -> (dx | None, dw, db | None); dy_add: a second gradient into the
    layer's output, summed inside the kernel; pool: a _ZeroPool the weight / bias gradients are taken from."""
    R, K = x2.shape
    O = dy2.shape[1]
    dx = torch.empty_like(x2) if want_dx else None
    dw = pool.take(w.shape) if pool is not None else torch.zeros_like(w)
    db = (pool.take((O,)) if pool is not None else torch.zeros(O, dtype=torch.float32, device=w.device)) if has_bias else None
    with torch.cuda.device(w.device):
        L.check(L.lib().cm_linear_act_backward(R, K, O, L.ptr(x2), L.ptr(w), layout, L.ptr(dy2), L.ptr(dy_add), L.ptr(y2), L.ptr(dx),
                                               L.ptr(dw), L.ptr(db), L.current_stream()), "cm_linear_act_backward")
    return dx, dw, db


def _fused_shape_ok(net, obs):
    """Shapes with a saved-forward instantiation (cm_*_forward_saved) and a backward chain."""
    # teams above 80 agents exceed the f16-split forward's LDS budget (its activation planes + the N x N score matrix):
    # they keep the per-layer path
    return (obs.is_cuda and 1 <= net._n_agents <= MAX_FUSED_AGENTS and 1 <= len(net.gcn_layers) <= 4
            and net._dec_obs_dim <= 96 and len(net.encoder._layers) == 1
            and os.environ.get("COMMARL_FUSED_TRAIN", "1") != "0" and os.environ.get("COMMARL_POLICY_KERNEL", "")[:1] not in ("f", "v"))


def _fused_train_ok(net, obs):
    return torch.is_grad_enabled() and _fused_shape_ok(net, obs)


class _FusedNetFn(torch.autograd.Function):
    """CommBaseNet trunk + head as ONE forward launch (cm_policy_forward_saved / cm_critic_forward_saved: the rollout
    kernel with stores of every activation the backward needs) and a hand-written backward chain:
    head layers <- residual <- L x (masked aggregation <- H.Wg) <- attention softmax <- linear_in <- encoder, each link one
    C-ABI kernel (cm_linear_act_backward, cm_masked_agg_backward, cm_attention_backward).  Returns logits [S,N,A]
    (policy) or per-agent values [S,N] (critic); the attention matrix comes back as a non-differentiable second output
    (the reference detaches nothing here, but no loss term reads it)."""

    @staticmethod
    def forward(ctx, net, obs, adj, ch, *params):
        N, Lh, d = net._n_agents, len(net.gcn_layers), net._dec_obs_dim
        S = obs.shape[0]
        R = S * N
        dev = obs.device
        policy = hasattr(net, "categorical_output_layer")
        z = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)   # noqa: E731
        obs2 = obs.reshape(R, d).contiguous()
        t = dict(a1=z(R, 128), e=z(R, 64), q=z(R, 64), hw=[z(R, 64) for _ in range(Lh)], h=[z(R, 64) for _ in range(Lh)])
        if policy:
            A = net._action_dim
            t.update(x1=z(R, 128), x2=z(R, 64), x3=z(R, 32), out=z(R, A))
        else:
            t.update(x1=z(R, 64), out=z(R))
        attn = z(S, N, N)
        sv = L.FwdSaves()
        sv.a1, sv.e, sv.q, sv.x1, sv.out = (t[k].data_ptr() for k in ("a1", "e", "q", "x1", "out"))
        if policy:
            sv.x2, sv.x3 = t["x2"].data_ptr(), t["x3"].data_ptr()
        for l in range(Lh):
            sv.hw[l], sv.h[l] = t["hw"][l].data_ptr(), t["h"][l].data_ptr()
        adj_c = None if adj is None else adj.contiguous()
        ch_c = None if ch is None else ch.contiguous()
        net._train_fwd = True                          # the weight pack refreshes the f16-split section only (_WeightPack._packed)
        # teams of 4, large batches: the wave-owned kernel of the rollout as training forward (cm_policy_forward_saved_wave: a
        # persistent workgroup per CU, activations in registers: 0.97 -> 0.5 ms at 1.1 M agent rows); its fragments are the
        # CM_PACK_WAVE section, refreshed together with the f16-split one then.  Small (launch-bound) batches keep the one section.
        wave = (N == 4 and S >= int(os.environ.get("COMMARL_TRAIN_FWD_WAVE_MIN", "16384"))
                and (os.environ.get("COMMARL_POLICY_KERNEL") or "w")[0] not in "hfv")
        net._train_fwd_wave = wave
        try:
            with torch.cuda.device(dev):
                if policy:
                    w = net._weights_struct()
                    rc = 1
                    if wave:
                        rc = L.lib().cm_policy_forward_saved_wave(C.byref(w), S, L.ptr(obs2), L.ptr(adj_c), L.ptr(ch_c), L.ptr(attn),
                                                                  C.byref(sv), L.current_stream())
                    if rc == 1:
                        rc = L.lib().cm_policy_forward_saved(C.byref(w), S, L.ptr(obs2), L.ptr(adj_c), L.ptr(ch_c), L.ptr(attn),
                                                             C.byref(sv), L.current_stream())
                else:
                    w = net._struct_from(net._packed())
                    w.mfma_pack = None if net._mfma is None else net._mfma.data_ptr()
                    vals = z(S)
                    rc = 1
                    if wave:
                        rc = L.lib().cm_critic_forward_saved_wave(C.byref(w), S, L.ptr(obs2), L.ptr(adj_c), L.ptr(ch_c), L.ptr(attn),
                                                                  L.ptr(vals), C.byref(sv), L.current_stream())
                    if rc == 1:
                        rc = L.lib().cm_critic_forward_saved(C.byref(w), S, L.ptr(obs2), L.ptr(adj_c), L.ptr(ch_c), L.ptr(attn),
                                                             L.ptr(vals), C.byref(sv), L.current_stream())
        finally:
            net._train_fwd = net._train_fwd_wave = False
        if rc == 1:
            raise L.CommarlError("no saved-forward instantiation for this shape (caller should have checked _fused_train_ok)")
        L.check(rc, "cm_*_forward_saved")
        ctx.net, ctx.t, ctx.obs2, ctx.adj, ctx.ch, ctx.attn, ctx.policy = net, t, obs2, adj_c, ch_c, attn, policy
        ctx.names = [n for n, _ in net.named_parameters()]
        ctx.mark_non_differentiable(attn)
        out = t["out"].view(S, N, -1) if policy else t["out"].view(S, N)
        return out, attn

    @staticmethod
    def backward(ctx, d_out, _d_attn):
        net, t, obs2, adj, ch, attn = ctx.net, ctx.t, ctx.obs2, ctx.adj, ctx.ch, ctx.attn
        N, Lh = net._n_agents, len(net.gcn_layers)
        R = obs2.shape[0]
        S = R // N
        P = dict(net.named_parameters())
        g = {}
        pool = _ZeroPool(list(P.values()), obs2.device, extra=31 * 64 * len(net.gcn_layers))   # (+ the bias-gradient replicas below)
        if ctx.policy:
            hd = net.categorical_output_layer
            lins = [l.linear for l in hd._layers] + [hd._output_layers[0].linear]
            pre = ["categorical_output_layer._layers.%d.linear" % i for i in range(3)] + ["categorical_output_layer._output_layers.0.linear"]
            acts = [t["h"][Lh - 1], t["x1"], t["x2"], t["x3"]]          # inputs of the four head layers
            d = d_out.reshape(R, -1).contiguous()
            for i in (3, 2, 1, 0):                                       # y of layer i = input of layer i + 1 (tanh), none for the logits
                d, g[pre[i] + ".weight"], g[pre[i] + ".bias"] = _lin_bwd(acts[i], lins[i].weight, 0, d, None if i == 3 else acts[i + 1],
                                                                          True, True, pool=pool)
        else:
            mm = net.baseline_aggregator._mean_module
            l1, l2 = mm._layers[0].linear, mm._output_layers[0].linear
            pre = "baseline_aggregator._mean_module."
            d = d_out.reshape(R, 1).contiguous()
            d, g[pre + "_output_layers.0.linear.weight"], g[pre + "_output_layers.0.linear.bias"] = _lin_bwd(t["x1"], l2.weight, 0, d, None, True, True, pool=pool)
            d, g[pre + "_layers.0.linear.weight"], g[pre + "_layers.0.linear.bias"] = _lin_bwd(t["h"][Lh - 1], l1.weight, 0, d, t["x1"], True, True, pool=pool)
        # d = gradient wrt the trunk output x = E + H_L (or H_L)
        e, q = t["e"], t["q"]
        d_res = d if net.residual else None                              # residual: the same gradient flows into E
        dH, d_attn, dhin0 = d, None, None
        with torch.cuda.device(obs2.device):
            for l in reversed(range(Lh)):
                gl = net.gcn_layers[l]
                minus = e if (l == Lh - 1 and net.residual) else None       # saved x = E + H_L: the hop's tanh output is x - E
                da, dhw = torch.empty_like(attn), torch.empty_like(e)
                # large batches of teams of 4: the bias gradient over 32 rows summed afterwards (cm_masked_agg_backward_r)
                reps = 32 if (N == 4 and S > 16 * 512) else 1
                dgb = pool.take((reps, 64)) if gl.bias is not None else None
                chan_ptr, stride = None, 0
                if ch is not None:
                    chan_ptr, stride = ch.data_ptr() + 4 * l * N * N, ch.shape[1] * N * N
                L.check(L.lib().cm_masked_agg_backward_r(S, N, 64, L.ptr(attn), L.ptr(adj), chan_ptr, stride, L.ptr(t["hw"][l]),
                                                         L.ptr(t["h"][l]), L.ptr(minus), L.ptr(dH), L.ptr(da), L.ptr(dhw), L.ptr(dgb),
                                                         reps, L.current_stream()), "cm_masked_agg_backward_r")
                if dgb is not None:
                    dgb = dgb.sum(0) if reps > 1 else dgb[0]
                d_attn = da if d_attn is None else d_attn.add_(da)
                hin = t["h"][l - 1] if l > 0 else e
                dhin, g["gcn_layers.%d.weight" % l], _ = _lin_bwd(hin, gl.weight, 1, dhw, None, True, False, pool=pool)
                if gl.bias is not None:
                    g["gcn_layers.%d.bias" % l] = dgb
                if l > 0:
                    dH = dhin
                else:
                    dhin0 = dhin
            # gradient wrt E = residual term + hop 0's input gradient + attention key side (summed by the attention kernel)
            # + linear_in's input gradient (summed by the encoder layer's backward): no accumulation passes
            if Lh == 0:
                raise L.CommarlError("fused training path needs at least one hop")
            dq, dE = torch.empty_like(q), torch.empty_like(e)
            L.check(L.lib().cm_attention_backward(S, N, 64, L.ptr(q), L.ptr(e), L.ptr(attn), L.ptr(d_attn), L.ptr(d_res), L.ptr(dhin0),
                                                  L.ptr(dq), L.ptr(dE), L.current_stream()), "cm_attention_backward")
        if net.attention_layer.attention_type == "general":
            deq, g["attention_layer.linear_in.weight"], _ = _lin_bwd(e, net.attention_layer.linear_in.weight, 0, dq, None, True, False, pool=pool)
        else:
            deq = dq                                                     # 'dot': Q is E itself
        enc1, enc2 = net.encoder._layers[0].linear, net.encoder._output_layers[0].linear
        # both encoder layers in one pass (the gradient wrt the hidden layer never leaves the workgroup); wide observations
        # (d > 64) take the two layers one by one
        dw2, db2 = pool.take(enc2.weight.shape), pool.take(enc2.bias.shape)
        dw1, db1 = pool.take(enc1.weight.shape), pool.take(enc1.bias.shape)
        with torch.cuda.device(obs2.device):
            rc = L.lib().cm_encoder_backward(R, obs2.shape[1], L.ptr(obs2), L.ptr(t["a1"]), L.ptr(e), L.ptr(enc2.weight), L.ptr(dE), L.ptr(deq),
                                             L.ptr(dw2), L.ptr(db2), L.ptr(dw1), L.ptr(db1), L.current_stream())
        if rc == 1:
            da1, dw2, db2 = _lin_bwd(t["a1"], enc2.weight, 0, dE, e, True, True, dy_add=deq)     # (dw2 .. db1 taken above stay zero, unused)
            _, dw1, db1 = _lin_bwd(obs2, enc1.weight, 0, da1, t["a1"], False, True)
        else:
            L.check(rc, "cm_encoder_backward")
        g["encoder._output_layers.0.linear.weight"], g["encoder._output_layers.0.linear.bias"] = dw2, db2
        g["encoder._layers.0.linear.weight"], g["encoder._layers.0.linear.bias"] = dw1, db1
        ctx.t = None                                                     # free the saved activations
        return (None, None, None, None) + tuple(g.get(n) for n in ctx.names)


@torch.no_grad()
def _fused_logits_nograd(net, obs, adj, ch, want_probs=False):
    """Policy logits [S,N,A] + attention from the saved-forward kernel with every store but the logits (and, on
    request, the action probabilities) switched off: loss / KL evaluations over the full batch, no activations kept.
    -> (logits, attn) or, with want_probs, (logits, probs)."""
    N, d, A = net._n_agents, net._dec_obs_dim, net._action_dim
    S = obs.shape[0]
    obs2 = obs.reshape(S * N, d).contiguous()
    out = torch.empty(S * N, A, dtype=torch.float32, device=obs.device)
    attn = torch.empty(S, N, N, dtype=torch.float32, device=obs.device)
    sv = L.FwdSaves()
    sv.out = out.data_ptr()
    probs = torch.empty(S, N, A, dtype=torch.float32, device=obs.device) if want_probs else None
    if want_probs:
        sv.probs = probs.data_ptr()
    adj_c = None if adj is None else adj.contiguous()
    ch_c = None if ch is None else ch.contiguous()
    # large batches of teams of 4: the same kernel the training forward takes (both sides of the PPO ratio from one arithmetic)
    wave = (N == 4 and S >= int(os.environ.get("COMMARL_TRAIN_FWD_WAVE_MIN", "16384"))
            and (os.environ.get("COMMARL_POLICY_KERNEL") or "w")[0] not in "hfv")
    with torch.cuda.device(obs.device):
        w = net._weights_struct()                            # (no-grad user: every section of the pack is current)
        rc = 1
        if wave:
            rc = L.lib().cm_policy_forward_saved_wave(C.byref(w), S, L.ptr(obs2), L.ptr(adj_c), L.ptr(ch_c), L.ptr(attn), C.byref(sv),
                                                      L.current_stream())
        if rc == 1:
            rc = L.lib().cm_policy_forward_saved(C.byref(w), S, L.ptr(obs2), L.ptr(adj_c), L.ptr(ch_c), L.ptr(attn), C.byref(sv),
                                                 L.current_stream())
    if rc == 1:
        raise L.CommarlError("no saved-forward instantiation for this shape")
    L.check(rc, "cm_policy_forward_saved")
    return (out.view(S, N, A), probs) if want_probs else (out.view(S, N, A), attn)


# ---------------------------------------------------------------------------------------------
# trunk
# ---------------------------------------------------------------------------------------------
def _as_dev(x, device):
    if x is None:
        return None
    if not torch.is_tensor(x):
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
    return x.to(device=device, dtype=torch.float32)


class _Stacked(list):
    """Equal-shape parameters that the flat weight copy stores back to back (what torch.stack would give, without its launch)."""

    def numel(self):
        return sum(t.numel() for t in self)


class _WeightPack:
    """Flat device copy of a net's (transposed) weights for the fused C-ABI kernels; subclasses list the
    tensors in ``_pack_tensors()``."""
    _pack_sig, _pack, _pack_stale, _train_fwd, _train_fwd_wave, _pack_fresh = None, None, False, False, False, 0

    def _pack_tensors(self):
        raise NotImplementedError

    def _packed(self):
        """Flat contiguous device copy of the (transposed) weights.  Allocated once; when a parameter
        changed (optimizer step / load_state_dict) the SAME buffer is rewritten in place, so device
        pointers - and any hipGraph that captured them - stay valid."""
        sig = tuple((p.data_ptr(), p._version) for p in self.parameters())
        # a training forward (_FusedNetFn sets _train_fwd around its launch) reads the f16-split fragments only: between two optimiser steps just that section is
        # refreshed (cm_*_pack_sections); the rest goes stale and is packed - with the range check - by the next no-grad user
        # (sync_weights() before a rollout, evaluate_nograd, act_device)
        partial = self._train_fwd and os.environ.get("COMMARL_PACK_PARTIAL", "1") != "0"
        need = (L.PACK_F16 | (L.PACK_WAVE if self._train_fwd_wave else 0)) if partial else L.PACK_ALL
        if sig == self._pack_sig and not (need & 7 & ~self._pack_fresh):     # every section this user reads is current
            return self._pack[1]
        with torch.no_grad():
            ts = self._pack_tensors()
            if self._pack is None or self._pack[0].device != next(self.parameters()).device:
                offs, off = {}, 0
                for k, v in ts.items():
                    if v is None:
                        offs[k] = None
                        continue
                    offs[k] = (off, v.numel())
                    off += (v.numel() + 3) & ~3                 # keep every tensor 16-byte aligned
                buf = torch.zeros(off, dtype=torch.float32, device=next(self.parameters()).device)
                base = buf.data_ptr()
                ptrs = {k: (None if o is None else base + 4 * o[0]) for k, o in offs.items()}
                self._pack = (buf, ptrs, offs)
            buf, ptrs, offs = self._pack
            self._flat_copy(ts, buf, offs)
            same = sig == self._pack_sig
            self._pack_sig = None                               # a refused pack (cm_*_pack: weight outside the f16 range) is retried
            self._after_pack(self._pack[1], need)
            self._pack_fresh = (need & 7) | (self._pack_fresh if same else 0)
            self._pack_sig, self._pack_stale = sig, self._pack_fresh != 7
        return self._pack[1]

    def _flat_copy(self, ts, buf, offs):
        """Every tensor of `ts` into its slot of the flat buffer.  On the GPU: ONE launch (cm_multi_copy_t) for all the plain
        parameters and transposed weight views (`weight.t()`: the source is the parameter itself, the kernel transposes; a
        _Stacked group entry by entry); anything else through the framework's copy."""
        plain = []
        for k, v in ts.items():
            if v is None:
                continue
            o, n = offs[k]
            if isinstance(v, _Stacked):                      # the stack's members one after the other
                parts, po = [], o
                for t in v:
                    parts.append((t, po, t.numel()))
                    po += t.numel()
            else:
                parts = [(v, o, n)]
            for v, o, n in parts:
                self._flat_copy_one(v, buf[o:o + n], n, plain, buf)
        if plain:
            m = len(plain)
            src = (C.c_void_p * m)(*[t[0] for t in plain])
            dstp = (C.c_void_p * m)(*[t[1] for t in plain])
            rows = (C.c_int32 * m)(*[t[2] for t in plain])
            cols = (C.c_int32 * m)(*[t[3] for t in plain])
            tr = (C.c_int32 * m)(*[t[4] for t in plain])
            with torch.cuda.device(buf.device):
                L.check(L.lib().cm_multi_copy_t(m, src, dstp, rows, cols, tr, L.current_stream()), "cm_multi_copy_t")

    @staticmethod
    def _flat_copy_one(v, dst, n, plain, buf):
        base = v._base if (v.dim() == 2 and v._base is not None and not v.is_contiguous()) else None
        if (buf.is_cuda and v.dtype == torch.float32 and v.dim() <= 2 and len(plain) < 40
                and (v.is_contiguous() or (base is not None and base.is_contiguous() and base.dim() == 2
                                           and base.shape == v.shape[::-1] and v.data_ptr() == base.data_ptr()))):
            if v.is_contiguous():
                plain.append((v.data_ptr(), dst.data_ptr(), 1, n, 0))
            else:                                            # v = base.t(): dst[c][r] = base[r][c]
                plain.append((base.data_ptr(), dst.data_ptr(), base.shape[0], base.shape[1], 1))
        else:
            dst.copy_(v.detach().to(torch.float32).reshape(-1))

    def _after_pack(self, ptrs, sections=15):
        """Hook: derived device-side layouts (the matrix-core operand pack) are rebuilt here."""

    def sync_weights(self):
        """Refresh the fused kernels' weight pack (call after an optimizer step, outside graphs)."""
        self._packed()

    def grad_norm(self):
        return float(np.sqrt(np.sum([p.grad.norm(2).item() ** 2 for p in self.parameters() if p.grad is not None])))

    def reset(self, dones=None):
        return

    @property
    def recurrent(self):
        return False


class CommBaseNet(_WeightPack, nn.Module):
    """comm_base_net.py:11-111."""
    _graph_capturable_update = True      # algos._UpdateGraphs: forward, loss and backward of these nets wait for nothing on the host

    def __init__(self, env_spec, n_agents, encoder_hidden_sizes=(128,), embedding_dim=64, attention_type="general",
                 n_gcn_layers=2, gcn_bias=True, state_include_actions=False, name="comm_base", residual=True,
                 device="cpu"):
        super().__init__()
        self.residual, self.device, self._n_agents, self.name = residual, device, n_agents, name
        self.comm = True
        self.centralized = True
        self.step = 0
        self.eps = 1e-12
        self._cent_obs_dim = env_spec.observation_space.flat_dim
        self._dec_obs_dim = int(self._cent_obs_dim / n_agents)
        self._action_dim = env_spec.action_space.n
        self._embedding_dim = embedding_dim
        self.n_gcn_layers = n_gcn_layers
        if state_include_actions:
            self._dec_obs_dim += self._action_dim
        self.encoder = MLPModule(self._dec_obs_dim, embedding_dim, encoder_hidden_sizes, output_tanh=True)
        self.attention_layer = AttentionModule(embedding_dim, attention_type)
        self.gcn_layers = nn.ModuleList([GraphConvolutionModule(embedding_dim, embedding_dim, bias=gcn_bias, id=i)
                                         for i in range(n_gcn_layers)])
        self._enc_hidden = tuple(encoder_hidden_sizes)

    # -- helpers --------------------------------------------------------------------------------
    def _flatten(self, obs_n, dist_adj, channels):
        """Reference reshapes (comm_categorical_mlp_policy.py:56-71): obs [...,N*d] -> [S,N,d],
        dist_adj [...,N*N]|[...,N,N] -> [S,N,N], channels [...,L*N,N]|[...,L,N,N] -> [S,L,N,N]."""
        N, Lh = self._n_agents, len(self.gcn_layers)
        lead = obs_n.shape[:-1] if obs_n.shape[-1] == N * self._dec_obs_dim else obs_n.shape[:-2]
        S = int(np.prod(lead)) if len(lead) else 1
        obs = obs_n.reshape(S, N, self._dec_obs_dim)
        adj = None if dist_adj is None else dist_adj.reshape(S, N, N)
        ch = None if channels is None else channels.reshape(S, Lh, N, N)
        return lead, S, obs, adj, ch

    def trunk(self, obs, adj, ch):
        """obs [S,N,d] -> (E, H_L, M) with autograd (CommBaseNet.forward :80-108)."""
        if not obs.is_cuda:
            raise L.CommarlError("policy / critic tensors must live on the MI355X (device cuda:k); there is no CPU path")
        E = self.encoder(obs)
        M = self.attention_layer(E)
        H = E
        for l, g in enumerate(self.gcn_layers):
            if _fused_ok(H, g.weight):
                hw = _LinearActFn.apply(H, g.weight, None, 0, 1)        # H.Wg, weight [in,out]
            else:
                hw = _MatmulWFn.apply(H, g.weight) if (H.is_cuda and torch.is_grad_enabled()) else torch.matmul(H, g.weight)
            H = masked_aggregate(M, adj, ch, l, hw, g.bias)
        return E, H, M

    # -- weight pack for the fused C-ABI forward -----------------------------------------------
    def _trunk_tensors(self):
        enc = self.encoder
        if len(enc._layers) != 1:
            raise L.CommarlError("fused forward is built for one encoder hidden layer (the runners' default)")
        t = OrderedDict()
        t["enc_w1t"] = enc._layers[0].linear.weight.t()
        t["enc_b1"] = enc._layers[0].linear.bias
        t["enc_w2t"] = enc._output_layers[0].linear.weight.t()
        t["enc_b2"] = enc._output_layers[0].linear.bias
        if self.attention_layer.attention_type == "general":
            t["attn_wt"] = self.attention_layer.linear_in.weight.t()
        else:                                   # 'dot': Q = E, i.e. the fused kernels' linear_in is the identity (a constant)
            t["attn_wt"] = torch.eye(self._embedding_dim, dtype=torch.float32, device=next(self.parameters()).device)
        t["gcn_w"] = _Stacked(g.weight for g in self.gcn_layers) if len(self.gcn_layers) else None
        t["gcn_b"] = (_Stacked(g.bias for g in self.gcn_layers)
                      if len(self.gcn_layers) and self.gcn_layers[0].bias is not None else None)
        return t

    def _head_tensors(self):
        raise NotImplementedError

    def _pack_tensors(self):
        ts = self._trunk_tensors()
        ts.update(self._head_tensors())
        return ts

    def _after_pack(self, ptrs, sections=15):
        """(Re)build the matrix-core operand pack (cm_policy_pack / cm_critic_pack; `sections`: which parts) in its own
        persistent buffer: same address for the life of the net, so captured hipGraphs keep reading fresh weights."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            return
        w = self._struct_from(ptrs)
        size_fn, pack_fn = (getattr(L.lib(), n) for n in self._mfma_fns)
        if self._mfma is None or self._mfma.device != dev:
            nbytes = size_fn(C.byref(w))
            self._mfma = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev) if nbytes else None
        if self._mfma is not None:
            with torch.cuda.device(dev):
                L.check(pack_fn(C.byref(w), L.ptr(self._mfma), int(sections), L.current_stream()), self._mfma_fns[1])


# ---------------------------------------------------------------------------------------------
# policy
# ---------------------------------------------------------------------------------------------
class CommCategoricalMLPPolicy(CommBaseNet):
    """comm_categorical_mlp_policy.py:8-141 (same ctor kwargs as runner_pp_commDP.py:49-61)."""

    def __init__(self, env_spec, n_agents, encoder_hidden_sizes=(128,), embedding_dim=64, attention_type="general",
                 n_gcn_layers=2, residual=True, gcn_bias=True, categorical_mlp_hidden_sizes=(128, 64, 32),
                 name="comm_categorical_mlp_policy", device="cpu"):
        super().__init__(env_spec=env_spec, n_agents=n_agents, encoder_hidden_sizes=encoder_hidden_sizes,
                         embedding_dim=embedding_dim, attention_type=attention_type, n_gcn_layers=n_gcn_layers,
                         gcn_bias=gcn_bias, name=name, device=device)
        self.residual = residual
        self._head_sizes = tuple(categorical_mlp_hidden_sizes)
        self.categorical_output_layer = MLPModule(embedding_dim, self._action_dim, categorical_mlp_hidden_sizes)
        self.seed, self.env_id_offset, self._policy_step = 1, 0, 0
        self.to(device)

    # -- autograd path (PPO update) --------------------------------------------------------------
    def _logits_flat(self, obs, adj, ch):
        """Raw head outputs [S,N,A] and attention [S,N,N] of flattened inputs: the fused training forward when it has an
        instantiation, under no_grad the same kernel storing only the logits, else the per-layer path."""
        fused_shape = len(self.categorical_output_layer._layers) == 3
        if _fused_train_ok(self, obs) and fused_shape:
            return _FusedNetFn.apply(self, obs, adj, ch, *self.parameters())          # one forward launch + hand-written backward
        if not torch.is_grad_enabled() and fused_shape and _fused_shape_ok(self, obs):
            return _fused_logits_nograd(self, obs, adj, ch)
        E, H, M = self.trunk(obs, adj, ch)
        x = E + H if self.residual else H
        return self.categorical_output_layer(x), M

    @torch.no_grad()
    def evaluate_nograd(self, obs_n, dist_adj, channels):
        """One no-grad forward over a batch -> (logits [...,N,A] or None, probs [...,N,A]): both from ONE launch when
        the fused training forward has an instantiation (the probabilities are the ones act_device returns, bit for bit),
        else the probabilities from act_device and no logits."""
        lead, S, obs, adj, ch = self._flatten(obs_n, dist_adj, channels)
        N = self._n_agents
        if _fused_shape_ok(self, obs) and len(self.categorical_output_layer._layers) == 3:
            logits, probs = _fused_logits_nograd(self, obs, adj, ch, want_probs=True)
            return logits.reshape(*lead, N, -1), probs.reshape(*lead, N, -1)
        _, probs, _ = self.act_device(obs.reshape(S, -1), None, adj, ch, want_actions=False, want_attn=False, policy_step=0)
        return None, probs.reshape(*lead, N, -1)

    def _logits(self, obs_n, dist_adj, channels):
        lead, S, obs, adj, ch = self._flatten(obs_n, dist_adj, channels)
        logits, _ = self._logits_flat(obs, adj, ch)
        return logits.reshape(*lead, self._n_agents, -1)

    def _probs(self, obs_n, avail_actions_n, dist_adj, channels):
        lead, S, obs, adj, ch = self._flatten(obs_n, dist_adj, channels)
        logits, M = self._logits_flat(obs, adj, ch)
        probs = torch.softmax(logits, dim=-1)
        if avail_actions_n is not None:
            probs = probs * avail_actions_n.reshape(S, self._n_agents, -1)
        probs = probs / probs.sum(dim=-1, keepdim=True)
        N = self._n_agents
        return probs.reshape(*lead, N, -1), M.reshape(*lead, N, N)

    def forward(self, obs_n, avail_actions_n, dist_adj, channels, get_actions=False):
        """-> (Categorical over [..., N, A], attention [..., N, N]).  get_actions=True takes
        numpy inputs and runs the fused no-grad kernel (reference :56-62,81-96)."""
        dev = next(self.parameters()).device
        if get_actions:
            obs_n, avail_actions_n = _as_dev(obs_n, dev), _as_dev(avail_actions_n, dev)
            dist_adj, channels = _as_dev(dist_adj, dev), _as_dev(channels, dev)
            _, probs, attn = self.act_device(obs_n, avail_actions_n, dist_adj, channels, want_actions=False)
            return Categorical(probs=probs.cpu()), attn.cpu()
        probs, attn = self._probs(obs_n, avail_actions_n, dist_adj, channels)
        return Categorical(probs=probs), attn

    def entropy(self, observations, avail_actions, dist_adj, channels):
        dists_n, _ = self.forward(observations, avail_actions, dist_adj, channels)
        return dists_n.entropy().mean(axis=-1)                       # :121-126

    def log_likelihood(self, observations, avail_actions, dist_adj, channels, actions):
        dists_n, _ = self.forward(observations, avail_actions, dist_adj, channels)
        return dists_n.log_prob(actions).sum(axis=-1)                # :128-137

    # -- fused rollout path ----------------------------------------------------------------------
    def _head_tensors(self):
        h = self.categorical_output_layer
        if len(h._layers) != 3:
            raise L.CommarlError("fused forward is built for the 3-hidden-layer categorical head (runner default)")
        t = OrderedDict()
        for i in range(3):
            t[f"hd_w{i + 1}t"] = h._layers[i].linear.weight.t()
            t[f"hd_b{i + 1}"] = h._layers[i].linear.bias
        t["hd_w4t"] = h._output_layers[0].linear.weight.t()
        t["hd_b4"] = h._output_layers[0].linear.bias
        return t

    def _struct_from(self, p):
        w = L.PolicyWeights()
        w.d, w.n_agents, w.n_hops = self._dec_obs_dim, self._n_agents, len(self.gcn_layers)
        w.enc_hidden, w.emb = self._enc_hidden[0], self._embedding_dim
        w.h1, w.h2, w.h3 = self._head_sizes
        w.n_act = self._action_dim
        w.no_residual = 0 if self.residual else 1
        for k, v in p.items():
            setattr(w, k, v)
        return w

    def _weights_struct(self):
        w = self._struct_from(self._packed())
        w.mfma_pack = None if self._mfma is None else self._mfma.data_ptr()
        return w

    _mfma, _mfma_fns = None, ("cm_policy_pack_bytes", "cm_policy_pack_sections")

    def set_rng(self, seed, env_id_offset=0):
        """Philox stream of the action sampler: counter (env id, policy step, site 7, agent)."""
        self.seed, self.env_id_offset = int(seed), int(env_id_offset)

    @torch.no_grad()
    def act_device(self, obs, avail, dist_adj, channels, greedy=False, want_actions=True, want_probs=True,
                   want_attn=True, out_actions=None, out_probs=None, out_attn=None, policy_step=None, step_base=None,
                   env_id_offset=None):
        """Fused forward on device tensors: obs [S,N*d]|[S,N,d]; avail/dist_adj/channels may be None
        (= all ones).  Returns (actions int32 [S,N], probs [S,N,A], attn [S,N,N]) as CUDA tensors."""
        dev = obs.device
        if dev.type != "cuda":
            raise L.CommarlError("the rollout forward is a HIP kernel: inputs must be CUDA tensors (no CPU fallback)")
        N, A = self._n_agents, self._action_dim
        S = obs.numel() // (N * self._dec_obs_dim)
        obs = obs.contiguous()
        actions = out_actions if out_actions is not None else (
            torch.empty(S, N, dtype=torch.int32, device=dev) if want_actions else None)
        probs = out_probs if out_probs is not None else (
            torch.empty(S, N, A, dtype=torch.float32, device=dev) if want_probs else None)
        attn = out_attn if out_attn is not None else (
            torch.empty(S, N, N, dtype=torch.float32, device=dev) if want_attn else None)
        if policy_step is None:
            policy_step = self._policy_step
            self._policy_step += 1
        if N > MAX_FUSED_AGENTS:
            return self._act_device_layers(obs, avail, dist_adj, channels, greedy, actions, probs, attn, policy_step, step_base,
                                           env_id_offset)
        w = self._weights_struct()
        with torch.cuda.device(dev):
            L.check(L.lib().cm_policy_forward(
                C.byref(w), S, L.ptr(obs), L.ptr(None if avail is None else avail.contiguous()),
                L.ptr(None if dist_adj is None else dist_adj.contiguous()),
                L.ptr(None if channels is None else channels.contiguous()), self.seed,
                self.env_id_offset if env_id_offset is None else int(env_id_offset),
                policy_step & 0xFFFFFFFF, L.ptr(step_base), int(greedy), L.ptr(actions), L.ptr(probs), L.ptr(attn),
                L.current_stream()), "cm_policy_forward")
        return actions, probs, attn

    def _act_device_layers(self, obs, avail, dist_adj, channels, greedy, actions, probs, attn, policy_step, step_base, env_id_offset):
        """Teams above 80 agents (PP map 40: N = 128; CO map 40: N = 96): one env's activation planes plus its N x N score matrix
        exceed a workgroup's 160 KB of LDS, so the forward runs layer by layer - encoder on the library GEMM, attention softmax
        and masked aggregation on their own HIP kernels (the training path's, up to 128 agents) - and the head + softmax x avail
        + Philox sample as ONE launch of the row-MLP kernel (cm_mlp_policy_forward) over x = E + H_L.  Same Philox site as the
        fused kernel: the sampled action is the oracle's inverse-CDF draw on the returned probabilities."""
        N, A, d = self._n_agents, self._action_dim, self._dec_obs_dim
        S = obs.numel() // (N * d)
        Lh = len(self.gcn_layers)
        E, H, M = self.trunk(obs.reshape(S, N, d), None if dist_adj is None else dist_adj.reshape(S, N, N),
                             None if channels is None else channels.reshape(S, Lh, N, N))
        x = (E + H if self.residual else H).reshape(S, N * self._embedding_dim).contiguous()
        hs = getattr(self, "_head_sampler_obj", None)
        if hs is None:
            hs = self.__dict__["_head_sampler_obj"] = _HeadSampler(self)       # (not a submodule: no new state_dict entries)
        hs.seed, hs.env_id_offset = self.seed, self.env_id_offset
        a_out, p_out, _ = hs.act_device(x, avail, greedy=greedy, want_actions=actions is not None, want_probs=True,
                                        out_actions=actions, out_probs=probs, policy_step=policy_step, step_base=step_base,
                                        env_id_offset=env_id_offset)
        if attn is not None:
            attn.copy_(M.reshape(attn.shape))
        return a_out, p_out, attn

    @torch.no_grad()
    def step_fused(self, env_batch, obs, dist_adj, channels, step_out, greedy=False, out_actions=None, out_probs=None,
                   out_attn=None, policy_step=0, step_base=None, env_id_offset=None, tape=None):
        """One sampler iteration in ONE launch (cm_rollout_step): this policy's forward + sample on `obs`, then the env
        step of `env_batch` on the sampled actions, results into `step_out` (an _lib.StepOut of device pointers).
        Returns False - having done nothing - when the library has no fused kernel for this shape."""
        w = self._weights_struct()
        with torch.cuda.device(obs.device):
            rc = L.lib().cm_rollout_step(
                env_batch._h, C.byref(w), L.ptr(obs), None, L.ptr(dist_adj), L.ptr(channels), self.seed,
                self.env_id_offset if env_id_offset is None else int(env_id_offset), policy_step & 0xFFFFFFFF,
                L.ptr(step_base), int(greedy), L.ptr(out_actions), L.ptr(out_probs), L.ptr(out_attn),
                C.byref(tape) if tape is not None else None, C.byref(step_out), L.current_stream())
        if rc == 1:
            return False
        L.check(rc, "cm_rollout_step")
        return True

    @torch.no_grad()
    def chunk_fused(self, env_batch, n_steps, strides, obs, dist_adj, channels, step_out, greedy=False, out_actions=None,
                    out_probs=None, out_attn=None, policy_step=0, step_base=None, env_id_offset=None, tail_next=None):
        """n_steps sampler iterations in ONE persistent launch (cm_rollout_chunk): pointers are those of the first
        step, `strides` (_lib.ChunkStrides) the per-step element strides of the time-major buffers.  tail_next = (obs,
        dist_adj | None, channels | None) of the slot that receives the last step's outputs, with the advance of `step_base`
        by n_steps (cm_rollout_chunk_tail: part of the same launch where the library can).  Returns False - having done
        nothing - when the library has no fused kernel for this shape."""
        w = self._weights_struct()
        eid = self.env_id_offset if env_id_offset is None else int(env_id_offset)
        with torch.cuda.device(obs.device):
            if tail_next is not None:
                rc = L.lib().cm_rollout_chunk_tail(
                    env_batch._h, C.byref(w), int(n_steps), C.byref(strides), L.ptr(obs), L.ptr(dist_adj), L.ptr(channels),
                    self.seed, eid, policy_step & 0xFFFFFFFF, L.ptr(step_base), int(greedy), L.ptr(out_actions), L.ptr(out_probs),
                    L.ptr(out_attn), C.byref(step_out), L.ptr(tail_next[0]), L.ptr(tail_next[1]), L.ptr(tail_next[2]),
                    L.current_stream())
            else:
                rc = L.lib().cm_rollout_chunk(
                    env_batch._h, C.byref(w), int(n_steps), C.byref(strides), L.ptr(obs), L.ptr(dist_adj), L.ptr(channels),
                    self.seed, eid, policy_step & 0xFFFFFFFF, L.ptr(step_base), int(greedy), L.ptr(out_actions), L.ptr(out_probs),
                    L.ptr(out_attn), C.byref(step_out), L.current_stream())
        if rc == 1:
            return False
        L.check(rc, "cm_rollout_chunk_tail" if tail_next is not None else "cm_rollout_chunk")
        return True

    def get_actions(self, obs_n, avail_actions_n, dist_adj, channels, greedy=False):
        """numpy in / numpy out, as the reference sampler calls it (:98-119)."""
        dev = next(self.parameters()).device
        obs = _as_dev(obs_n, dev)
        S = obs.shape[0]
        act, probs, attn = self.act_device(obs.reshape(S, -1), _as_dev(avail_actions_n, dev), _as_dev(dist_adj, dev),
                                           _as_dev(channels, dev), greedy=greedy)
        probs, attn = probs.cpu().numpy(), attn.cpu().numpy()
        infos = dict(action_probs=[probs[i] for i in range(S)], attention_weights=[attn[i] for i in range(S)])
        return act.cpu().numpy().astype(np.int64), infos


# ---------------------------------------------------------------------------------------------
# critic
# ---------------------------------------------------------------------------------------------
class GaussianMLPModule(nn.Module):
    """gaussian_mlp_module.py:62-188 in the configuration the critic uses: learned shared
    log-std ``_init_std`` (init log 1.0), min_std 1e-6, exp parameterisation."""

    def __init__(self, input_dim, output_dim, hidden_sizes=(32, 32), init_std=1.0, min_std=1e-6):
        super().__init__()
        self._init_std = nn.Parameter(torch.Tensor([init_std]).log())
        self._min_std_param = None if min_std is None else math.log(min_std)
        self._mean_module = MLPModule(input_dim, output_dim, hidden_sizes)

    def forward(self, x):
        mean = self._mean_module(x)
        ls = self._init_std if self._min_std_param is None else self._init_std.clamp(min=self._min_std_param)
        return mean, ls.exp()


class _GaussNLLFn(torch.autograd.Function):
    """The critic's Gaussian NLL from the per-agent outputs in one launch, its gradient in one more (cm_gauss_nll_*:
    comm_base_critic.py:59-89 + :110-112 + the std of gaussian_mlp_module.py) instead of ~25 framework launches each way."""

    @staticmethod
    def forward(ctx, per_agent, returns, log_std, min_log_std, ws):
        S, N = per_agent.shape
        pa, r = per_agent.contiguous(), returns.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=pa.device)
        has_min = 0 if min_log_std is None else 1
        with torch.cuda.device(pa.device):
            L.check(L.lib().cm_gauss_nll_forward(S, N, L.ptr(pa), L.ptr(r), L.ptr(log_std), float(min_log_std or 0.0), has_min, L.ptr(out),
                                                 L.ptr(ws), L.current_stream()), "cm_gauss_nll_forward")
        ctx.save_for_backward(pa, r, log_std, out)
        ctx.min_log_std = min_log_std
        return out[0]

    @staticmethod
    def backward(ctx, g):
        pa, r, log_std, out = ctx.saved_tensors
        S, N = pa.shape
        d_pa, d_ls = torch.empty_like(pa), torch.empty_like(log_std)
        g = g.to(torch.float32).contiguous()
        with torch.cuda.device(pa.device):
            L.check(L.lib().cm_gauss_nll_backward(S, N, L.ptr(pa), L.ptr(r), L.ptr(log_std), float(ctx.min_log_std or 0.0),
                                                  0 if ctx.min_log_std is None else 1, L.ptr(out), L.ptr(g), L.ptr(d_pa), L.ptr(d_ls),
                                                  L.current_stream()), "cm_gauss_nll_backward")
        return d_pa, None, d_ls, None, None


class CommBaseCritic(CommBaseNet):
    """comm_base_critic.py:11-122, aggregators 'sum' and 'direct' (same ctor kwargs as runner_pp_commDP.py:63-74)."""

    def __init__(self, env_spec, n_agents, encoder_hidden_sizes=(128,), embedding_dim=64, decoder_hidden_sizes=(64,),
                 attention_type="general", n_gcn_layers=2, residual=True, gcn_bias=True, share_std=False,
                 state_include_actions=False, aggregator_type="sum", name="base_critic", device="cpu"):
        super().__init__(env_spec=env_spec, n_agents=n_agents, encoder_hidden_sizes=encoder_hidden_sizes,
                         embedding_dim=embedding_dim, attention_type=attention_type, n_gcn_layers=n_gcn_layers,
                         residual=residual, gcn_bias=gcn_bias, state_include_actions=state_include_actions, name=name,
                         device=device)
        if aggregator_type not in ("sum", "direct"):
            raise ValueError("aggregator_type must be 'sum' or 'direct' (comm_base_critic.py:46-49)")
        self.aggregator_type = aggregator_type
        if aggregator_type != "sum" or n_agents > MAX_FUSED_AGENTS:
            self._graph_capturable_update = False        # framework GEMMs / per-layer fallbacks in the step: eager
        self._dec_hidden = tuple(decoder_hidden_sizes)
        # 'sum': one value per agent from its embedding, summed (:110-112).  'direct': ONE value from the concatenated
        # embeddings of the whole team (:113-116) - the head is then a plain [N * 64] -> 64 -> 1 MLP on the framework's GEMMs
        # behind the fused trunk kernels (per-layer path; the fused critic kernels carry the per-agent head only)
        agg_in = embedding_dim if aggregator_type == "sum" else embedding_dim * n_agents
        self.baseline_aggregator = GaussianMLPModule(agg_in, 1, hidden_sizes=decoder_hidden_sizes)
        self.to(device)

    def sync_weights(self):
        if self.aggregator_type == "sum":
            super().sync_weights()                       # 'direct' has no fused kernel, hence no weight pack to refresh

    def _head_tensors(self):
        m = self.baseline_aggregator._mean_module
        if len(m._layers) != 1:
            raise L.CommarlError("fused critic forward is built for one decoder hidden layer (default)")
        return OrderedDict(dec_w1t=m._layers[0].linear.weight.t(), dec_b1=m._layers[0].linear.bias,
                           dec_w2t=m._output_layers[0].linear.weight.t(), dec_b2=m._output_layers[0].linear.bias)

    _mfma, _mfma_fns = None, ("cm_critic_pack_bytes", "cm_critic_pack_sections")

    def _struct_from(self, p):
        w = L.CriticWeights()
        w.d, w.n_agents, w.n_hops = self._dec_obs_dim, self._n_agents, len(self.gcn_layers)
        w.enc_hidden, w.emb, w.dec_hidden = self._enc_hidden[0], self._embedding_dim, self._dec_hidden[0]
        w.no_residual = 0 if self.residual else 1
        for k, v in p.items():
            setattr(w, k, v)
        return w

    def _fused_head_ok(self, obs):
        return (self.aggregator_type == "sum" and _fused_train_ok(self, obs)
                and len(self.baseline_aggregator._mean_module._layers) == 1)

    def _values_grad(self, obs_n, dist_adj, channels):
        lead, S, obs, adj, ch = self._flatten(obs_n, dist_adj, channels)
        if self.aggregator_type == "direct":
            E, H, _ = self.trunk(obs, adj, ch)
            x = E + H if self.residual else H
            mean, std = self.baseline_aggregator(x.reshape(S, -1))                   # concatenated embeddings (:113-115)
            return mean.squeeze(-1).reshape(*lead), std
        if self._fused_head_ok(obs):
            per_agent, _ = _FusedNetFn.apply(self, obs, adj, ch, *self.parameters())  # [S,N] per-agent means
            ag = self.baseline_aggregator
            ls = ag._init_std if ag._min_std_param is None else ag._init_std.clamp(min=ag._min_std_param)
            return per_agent.sum(-1).reshape(*lead), ls.exp()
        E, H, _ = self.trunk(obs, adj, ch)
        x = E + H if self.residual else H
        mean, std = self.baseline_aggregator(x)
        return mean.squeeze(-1).sum(-1).reshape(*lead), std

    @torch.no_grad()
    def values_device(self, obs, dist_adj, channels, out=None):
        """Fused no-grad forward (cm_critic_forward): obs [..., N*d] CUDA -> values [...]."""
        dev = obs.device
        if dev.type != "cuda":
            raise L.CommarlError("critic forward is a HIP kernel: inputs must be CUDA tensors (no CPU fallback)")
        if self.aggregator_type == "direct" or self._n_agents > MAX_FUSED_AGENTS:    # no one-launch kernel: layer by layer
            v = self._values_grad(obs, dist_adj, channels)[0]
            return v if out is None else out.copy_(v.reshape(out.shape))
        N = self._n_agents
        lead = obs.shape[:-1] if obs.shape[-1] == N * self._dec_obs_dim else obs.shape[:-2]
        S = obs.numel() // (N * self._dec_obs_dim)
        values = out if out is not None else torch.empty(S, dtype=torch.float32, device=dev)
        w = self._struct_from(self._packed())
        w.mfma_pack = None if self._mfma is None else self._mfma.data_ptr()
        with torch.cuda.device(dev):
            L.check(L.lib().cm_critic_forward(C.byref(w), S, L.ptr(obs.contiguous()),
                                              L.ptr(None if dist_adj is None else dist_adj.contiguous()),
                                              L.ptr(None if channels is None else channels.contiguous()),
                                              L.ptr(values), L.current_stream()), "cm_critic_forward")
        return values.reshape(*lead) if out is None else values

    def forward(self, obs_n, avail_actions_n, dist_adj, channels, get_actions=False):
        """-> values [P,T] (:91-114).  Under no_grad this is the fused kernel."""
        dev = next(self.parameters()).device
        if get_actions or not torch.is_tensor(obs_n):
            obs_n, dist_adj, channels = _as_dev(obs_n, dev), _as_dev(dist_adj, dev), _as_dev(channels, dev)
        if not torch.is_grad_enabled():
            return self.values_device(obs_n, dist_adj, channels)
        return self._values_grad(obs_n, dist_adj, channels)[0]

    def compute_loss(self, obs_n, returns, dist_adj, channels, get_actions=False):
        """Gaussian NLL with the shared learned std; padded steps are included in the mean (:59-89)."""
        lead, S, obs, adj, ch = self._flatten(obs_n, dist_adj, channels)
        ag = self.baseline_aggregator
        if (self._fused_head_ok(obs) and torch.is_tensor(returns) and returns.is_cuda and returns.dtype == torch.float32
                and ag._init_std.numel() == 1 and os.environ.get("COMMARL_FUSED_NLL", "1") != "0"):
            per_agent, _ = _FusedNetFn.apply(self, obs, adj, ch, *self.parameters())  # [S,N] per-agent means
            ws = getattr(self, "_nll_ws", None)
            if ws is None or ws.device != per_agent.device:
                ws = self._nll_ws = torch.zeros(2, dtype=torch.float64, device=per_agent.device)   # CM_GAUSS_WS_BYTES, left zero by each launch
            return _GaussNLLFn.apply(per_agent, returns.reshape(-1), ag._init_std, ag._min_std_param, ws)
        values, std = self._values_grad(obs_n, dist_adj, channels)
        # validate_args=False: the constructor's argument check reads a device flag back (a host sync per optimiser step, and not
        # capturable into the update's hipGraphs); a non-finite value still surfaces - as a NaN loss in the epoch's statistics
        return -Normal(values, std.mean(), validate_args=False).log_prob(returns).mean()


# ---------------------------------------------------------------------------------------------
# Obs-DP / CENT variants: no communication, plain row-wise MLPs
# ---------------------------------------------------------------------------------------------
class _RowMLPPolicy(_WeightPack):
    """Shared rollout path of the two non-communicating policies: one fused HIP launch
    (cm_mlp_policy_forward, csrc/cm_mlp.hip) over the layer chain listed by ``_chain()``."""

    def _chain(self):
        """-> list of (nn.Linear, tanh?) in forward order."""
        raise NotImplementedError

    def _pack_tensors(self):
        t = OrderedDict()
        for i, (lin, _) in enumerate(self._chain()):
            t[f"w{i}t"] = lin.weight.t()
            t[f"b{i}"] = lin.bias
        return t

    _mlp_pack = None

    def _after_pack(self, ptrs, sections=15):
        """(Re)build the B-fragment pack of the chain (cm_mlp_pack) in a persistent buffer."""
        dev = next(self.parameters()).device
        if dev.type != "cuda":
            return
        w = self._struct_from(ptrs)
        if self._mlp_pack is None or self._mlp_pack.device != dev:
            self._mlp_pack = torch.zeros(L.lib().cm_mlp_pack_bytes(C.byref(w)) // 4, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            L.check(L.lib().cm_mlp_pack(C.byref(w), L.ptr(self._mlp_pack), L.current_stream()), "cm_mlp_pack")

    def _mlp_struct(self):
        w = self._struct_from(self._packed())
        w.mfma_pack = None if self._mlp_pack is None else self._mlp_pack.data_ptr()
        return w

    def _struct_from(self, p):
        chain = self._chain()
        if len(chain) > L.MLP_MAX_LAYERS:
            raise L.CommarlError(f"fused MLP forward takes at most {L.MLP_MAX_LAYERS} linear layers")
        w = L.MlpWeights()
        w.in_dim, w.n_layers, w.tanh_mask = chain[0][0].in_features, len(chain), 0
        for i, (lin, th) in enumerate(chain):
            w.out_dim[i] = lin.out_features
            w.wt[i], w.b[i] = p[f"w{i}t"], p[f"b{i}"]
            w.tanh_mask |= int(bool(th)) << i
        return w

    def set_rng(self, seed, env_id_offset=0):
        """Philox stream of the action sampler: counter (env id, policy step, site 7, agent)."""
        self.seed, self.env_id_offset = int(seed), int(env_id_offset)

    @torch.no_grad()
    def act_device(self, obs, avail=None, dist_adj=None, channels=None, greedy=False, want_actions=True,
                   want_probs=True, want_attn=False, out_actions=None, out_probs=None, out_attn=None,
                   policy_step=None, step_base=None, env_id_offset=None):
        """obs [S,N*d] (CUDA) -> (actions int32 [S,N], probs [S,N,A], None).  dist_adj / channels are accepted
        and ignored so the rollout engine drives every policy through one call shape."""
        dev = obs.device
        if dev.type != "cuda":
            raise L.CommarlError("the rollout forward is a HIP kernel: inputs must be CUDA tensors (no CPU fallback)")
        N, A = self._n_agents, self._action_dim
        S = obs.numel() // (N * self._dec_obs_dim)
        obs = obs.contiguous()
        actions = out_actions if out_actions is not None else (
            torch.empty(S, N, dtype=torch.int32, device=dev) if want_actions else None)
        probs = out_probs if out_probs is not None else (
            torch.empty(S, N, A, dtype=torch.float32, device=dev) if want_probs else None)
        if policy_step is None:
            policy_step = self._policy_step
            self._policy_step += 1
        rows, groups = (S * N, 1) if self._per_agent_rows else (S, N)
        w = self._mlp_struct()
        with torch.cuda.device(dev):
            L.check(L.lib().cm_mlp_policy_forward(
                C.byref(w), rows, groups, A, N, L.ptr(obs), L.ptr(None if avail is None else avail.contiguous()),
                self.seed, self.env_id_offset if env_id_offset is None else int(env_id_offset),
                policy_step & 0xFFFFFFFF, L.ptr(step_base), int(greedy), L.ptr(actions), L.ptr(probs),
                L.current_stream()), "cm_mlp_policy_forward")
        return actions, probs, None

    def get_actions(self, observations, avail_actions, greedy=False):
        """numpy in / numpy out (dec_categorical_mlp_policy.py:150-176, centralized_...:99-118)."""
        dev = next(self.parameters()).device
        obs = _as_dev(observations, dev)
        S = obs.shape[0]
        act, probs, _ = self.act_device(obs.reshape(S, -1), _as_dev(avail_actions, dev), greedy=greedy)
        probs = probs.cpu().numpy()
        return act.cpu().numpy().astype(np.int64), dict(action_probs=[probs[i] for i in range(S)])

    def forward(self, obs_n, avail_actions_n, get_actions=False):
        """-> Categorical over [..., N, A] (masked + renormalised)."""
        dev = next(self.parameters()).device
        if get_actions:
            obs_n, avail_actions_n = _as_dev(obs_n, dev), _as_dev(avail_actions_n, dev)
            _, probs, _ = self.act_device(obs_n.reshape(-1, self._n_agents * self._dec_obs_dim), avail_actions_n,
                                          want_actions=False)
            return Categorical(probs=probs.reshape(*obs_n.shape[:-1], self._n_agents, -1).cpu())
        return Categorical(probs=self._probs(obs_n, avail_actions_n)[0])

    @staticmethod
    def _need_gpu(x):
        if not x.is_cuda:
            raise L.CommarlError("policy tensors must live on the MI355X (device cuda:k); there is no CPU path")

    def _masked(self, logits, avail_actions_n):
        probs = torch.softmax(logits, dim=-1)
        if avail_actions_n is not None:
            probs = probs * avail_actions_n.reshape(probs.shape)
        return probs / probs.sum(dim=-1, keepdim=True)

    def entropy(self, observations, avail_actions):
        return self.forward(observations, avail_actions).entropy().mean(axis=-1)

    def log_likelihood(self, observations, avail_actions, actions):
        return self.forward(observations, avail_actions).log_prob(actions).sum(axis=-1)

    @property
    def vectorized(self):
        return True


class _HeadSampler(_RowMLPPolicy):
    """The Comm-DP policy's head (64 -> 128 -> 64 -> 32 -> A, categorical_mlp_module.py:64-80) + softmax x avail + Philox sample as
    a row-MLP chain over the trunk output x = E + H_L: the rollout forward of teams too large for the one-launch kernel."""
    _per_agent_rows = True

    def __init__(self, policy):
        self._policy = policy
        self._n_agents, self._action_dim = policy._n_agents, policy._action_dim
        self._dec_obs_dim = self._obs_dim = policy._embedding_dim
        self.seed, self.env_id_offset, self._policy_step = policy.seed, policy.env_id_offset, 0

    def parameters(self):
        return self._policy.categorical_output_layer.parameters()

    def _chain(self):
        h = self._policy.categorical_output_layer
        return [(l.linear, True) for l in h._layers] + [(h._output_layers[0].linear, False)]


class DecCategoricalMLPPolicy(_RowMLPPolicy, MLPModule):
    """dec_categorical_mlp_policy.py:14-232 (ctor of runner_pp_obsDP.py:52-59).  ``hidden_sizes`` =
    (encoder hidden, embedding, head hidden); parameters: ``_layers.0`` / ``_output_layers.0`` (head, created
    first as in the reference) and ``encoder.*``."""
    _per_agent_rows = True

    def __init__(self, env_spec, n_agents, hidden_sizes=(32, 32), name="DecCategoricalMLPPolicy", device="cpu",
                 _embedding_dim=64, **unused):
        self._n_agents = n_agents
        self._dec_obs_dim = self._obs_dim = int(env_spec.observation_space.flat_dim / n_agents)
        self._action_dim = env_spec.action_space.n
        self._embedding_dim = hidden_sizes[1]
        MLPModule.__init__(self, self._embedding_dim, self._action_dim, (hidden_sizes[-1],))
        self.encoder = MLPModule(self._obs_dim, self._embedding_dim, (hidden_sizes[0],), output_tanh=True)
        self.device, self.name, self.step, self.centralized = device, name, 0, True
        self.seed, self.env_id_offset, self._policy_step = 1, 0, 0
        self.to(device)

    def _chain(self):
        e = self.encoder
        return ([(l.linear, True) for l in e._layers] + [(e._output_layers[0].linear, True)]
                + [(l.linear, True) for l in self._layers] + [(self._output_layers[0].linear, False)])

    def _logits(self, obs_n, dist_adj=None, channels=None):
        """Raw per-agent logits [..., N, A] (what CentralizedMAPPO's one-launch surrogate loss consumes)."""
        self._need_gpu(obs_n)
        obs = obs_n.reshape(obs_n.shape[:-1] + (self._n_agents, -1))              # :110
        return MLPModule.forward(self, self.encoder(obs))

    def _probs(self, obs_n, avail_actions_n, dist_adj=None, channels=None):
        return self._masked(self._logits(obs_n), avail_actions_n), None


class CentralizedCategoricalMLPPolicy(_RowMLPPolicy, MLPModule):
    """centralized_categorical_mlp_policy.py:11-144 (ctor of runner_pp_cent.py:51-59): one MLP over the
    concatenated observation, N x 5 logits, agents' actions independent given the joint observation."""
    _per_agent_rows = False

    def __init__(self, env_spec, n_agents, hidden_sizes=(32, 32), name="CentralizedCategoricalMLPPolicy", device="cpu",
                 **unused):
        self._n_agents = n_agents
        self._obs_dim = env_spec.observation_space.flat_dim
        self._dec_obs_dim = self._obs_dim // n_agents
        self._action_dim = env_spec.action_space.n
        MLPModule.__init__(self, self._obs_dim, self._action_dim * n_agents, tuple(hidden_sizes))
        self.device, self.name, self.step, self.centralized = device, name, 0, True
        self.seed, self.env_id_offset, self._policy_step = 1, 0, 0
        self.to(device)

    def _chain(self):
        return [(l.linear, True) for l in self._layers] + [(self._output_layers[0].linear, False)]

    def _logits(self, obs_n, dist_adj=None, channels=None):
        """Raw logits [..., N, A] (what CentralizedMAPPO's one-launch surrogate loss consumes)."""
        self._need_gpu(obs_n)
        logits = MLPModule.forward(self, obs_n)
        return logits.reshape(logits.shape[:-1] + (self._n_agents, -1))           # :83

    def _probs(self, obs_n, avail_actions_n, dist_adj=None, channels=None):
        return self._masked(self._logits(obs_n), avail_actions_n), None


class GaussianMLPBaseline(_WeightPack, nn.Module):
    """com_marl/torch/baselines/gaussian_mlp_baseline.py:7-115 (runner_pp_cent.py:61-63): V(s) from the
    concatenated observation, Gaussian NLL with a learned shared std (no clamp: min_std=None)."""

    def __init__(self, env_spec, hidden_sizes=(32, 32), learn_std=True, init_std=1.0, name="GaussianMLPBaseline",
                 device="cpu", **unused):
        super().__init__()
        self.name = name
        self.input_dim = env_spec.observation_space.flat_dim
        self.module = GaussianMLPModule(self.input_dim, 1, hidden_sizes=tuple(hidden_sizes), init_std=init_std,
                                        min_std=None)
        if not learn_std:
            self.module._init_std.requires_grad_(False)
        self.device = device
        self.to(device)

    def _chain(self):
        m = self.module._mean_module
        return [(l.linear, True) for l in m._layers] + [(m._output_layers[0].linear, False)]

    _pack_tensors = _RowMLPPolicy._pack_tensors
    _mlp_struct = _RowMLPPolicy._mlp_struct
    _struct_from = _RowMLPPolicy._struct_from
    _after_pack = _RowMLPPolicy._after_pack
    _mlp_pack = None

    @torch.no_grad()
    def values_device(self, obs):
        dev = obs.device
        if dev.type != "cuda":
            raise L.CommarlError("baseline forward is a HIP kernel: inputs must be CUDA tensors (no CPU fallback)")
        lead = obs.shape[:-1]
        rows = obs.numel() // self.input_dim
        values = torch.empty(rows, dtype=torch.float32, device=dev)
        w = self._mlp_struct()
        with torch.cuda.device(dev):
            L.check(L.lib().cm_mlp_value_forward(C.byref(w), rows, L.ptr(obs.contiguous()), L.ptr(values),
                                                 L.current_stream()), "cm_mlp_value_forward")
        return values.reshape(*lead)

    def forward(self, obs):
        """-> values [P,T] (:100-115).  Under no_grad this is the fused kernel."""
        if not torch.is_tensor(obs):
            obs = _as_dev(obs, next(self.parameters()).device)
        if not torch.is_grad_enabled():
            return self.values_device(obs)
        return self.module(obs)[0].flatten(-2)

    def compute_loss(self, obs, returns):
        _RowMLPPolicy._need_gpu(obs)
        mean, std = self.module(obs.reshape(-1, self.input_dim))                  # :93-96
        return -Normal(mean, std, validate_args=False).log_prob(returns.reshape(-1, 1)).mean()      # (no host sync, as above)
