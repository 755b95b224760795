"""GPU: the sequence-persistent LSTM kernels (csrc/lstm.hip) against a CPU evaluation in fp64 of torch.nn.LSTM run
step by step with the reference's done-reset semantics (rl_games/common/layers/recurrent.py:26-58; host mirror
policy.RnnWithDones), forward and - through autograd on the CPU side - backward.  (Round 2 compared with
torch.nn.LSTM on the same GPU, i.e. with MIOpen; the reference here is independent of any GPU library.)

Tolerance: the kernels are fp32, the reference exact to fp32 resolution: rtol 1e-5 north_star tolerance on O(1)
activations plus an absolute term for values near zero; gradients are compared relative to the tensor scale."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _reference(x, lstm, h0, c0, dones, T):
    """x [S*T, I] rows (seq, t).  Returns out [S*T, H], (hT, cT)."""
    S = x.shape[0] // T
    xs = x.reshape(S, T, -1).transpose(0, 1)
    d = dones.reshape(S, T).t() if dones is not None else None
    st = (h0.unsqueeze(0), c0.unsqueeze(0))
    outs = []
    for t in range(T):
        if d is not None:
            keep = (1.0 - d[t].float()).reshape(1, -1, 1)
            st = (st[0] * keep, st[1] * keep)
        o, st = lstm(xs[t:t + 1], st)
        outs.append(o)
    out = torch.cat(outs, 0).transpose(0, 1).reshape(S * T, -1)
    return out, st


@pytest.mark.parametrize('S,T,I,H,with_dones', [(64, 16, 64, 64, True), (37, 4, 12, 32, True),
                                              (1024, 16, 64, 64, True), (5, 1, 7, 16, False),
                                              (130, 8, 20, 64, False)])
def test_lstm_forward_backward_match_torch(S, T, I, H, with_dones):
    from rl_games_amd import ops
    g = torch.Generator().manual_seed(S * 7 + T)
    lstm32 = torch.nn.LSTM(I, H, 1)
    x32 = torch.randn(S * T, I, generator=g)
    h0 = (0.5 * torch.randn(S, H, generator=g)).to(DEV)
    c0 = (0.5 * torch.randn(S, H, generator=g)).to(DEV)
    dones = (torch.rand(S * T, generator=g) < 0.2).to(torch.uint8).to(DEV) if with_dones else None
    d_out = torch.randn(S * T, H, generator=g).to(DEV)

    # the reference: CPU, fp64, the fp32 parameters and inputs upcast exactly
    lstm = torch.nn.LSTM(I, H, 1).double()
    lstm.load_state_dict({k: v.double() for k, v in lstm32.state_dict().items()})
    x = x32.double().requires_grad_(True)
    ref_out, (ref_h, ref_c) = _reference(x, lstm, h0.cpu().double(), c0.cpu().double(),
                                         None if dones is None else dones.cpu(), T)
    ref_out.backward(d_out.cpu().double())
    ref_out, ref_h, ref_c = ref_out.float().to(DEV), ref_h.float().to(DEV), ref_c.float().to(DEV)

    class _Grads:
        pass
    for name in ('weight_ih_l0', 'weight_hh_l0', 'bias_ih_l0', 'bias_hh_l0'):
        t = _Grads()
        t.grad = getattr(lstm, name).grad.float().to(DEV)
        setattr(lstm, '_' + name, t)
    x_grad = x.grad.float().to(DEV)
    x = x32.to(DEV)

    w_ih, w_hh = lstm32.weight_ih_l0.detach().to(DEV), lstm32.weight_hh_l0.detach().to(DEV)
    bias = (lstm32.bias_ih_l0 + lstm32.bias_hh_l0).detach().to(DEV)
    gates = torch.addmm(bias, x.detach(), w_ih.t())
    out = torch.empty(S * T, H, device=DEV)
    c_all = torch.empty(S * T, H, device=DEV)
    hprev = torch.empty(S * T, H, device=DEV)
    hT, cT = torch.empty(S, H, device=DEV), torch.empty(S, H, device=DEV)
    ops.lstm_seq_forward(gates, w_hh.contiguous(), h0, c0, dones, out, c_all, hprev, hT, cT, seq_len=T)
    assert torch.allclose(out, ref_out.detach(), rtol=1e-5, atol=2e-6)
    assert torch.allclose(hT, ref_h[0].detach(), rtol=1e-5, atol=2e-6)
    assert torch.allclose(cT, ref_c[0].detach(), rtol=1e-5, atol=2e-6)

    d_gates = torch.empty(S * T, 4 * H, device=DEV)
    ops.lstm_seq_backward(gates, c_all, c0, dones, w_hh.contiguous(), d_out, d_gates, T)
    dx = d_gates @ w_ih
    dw_ih = d_gates.t() @ x.detach()
    dw_hh = d_gates.t() @ hprev
    db = d_gates.sum(0)

    def close(a, b, name):
        scale = b.abs().max().item()
        assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-7, (name, (a - b).abs().max().item(), scale)
    close(dx, x_grad, 'dx')
    close(dw_ih, lstm._weight_ih_l0.grad, 'dw_ih')
    close(dw_hh, lstm._weight_hh_l0.grad, 'dw_hh')
    close(db, lstm._bias_ih_l0.grad, 'db_ih')
    close(db, lstm._bias_hh_l0.grad, 'db_hh')


def test_lstm_rejects_unsupported_hidden_and_cpu():
    from rl_games_amd import ops
    assert ops.lstm_supported(64) and not ops.lstm_supported(100)
    g = torch.zeros(4, 400, device=DEV)
    with pytest.raises(RuntimeError):
        ops.lstm_seq_forward(g, torch.zeros(400, 100, device=DEV), torch.zeros(4, 100, device=DEV),
                             torch.zeros(4, 100, device=DEV), None, torch.zeros(4, 100, device=DEV))
    with pytest.raises(Exception):
        ops.lstm_seq_forward(torch.zeros(4, 256), torch.zeros(256, 64), torch.zeros(4, 64), torch.zeros(4, 64),
                             None, torch.zeros(4, 64))
