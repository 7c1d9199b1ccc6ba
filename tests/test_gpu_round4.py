"""-m gpu, round 4: what was added this round and is not covered where its neighbours live.
  * `model.precision: high` with a BiLSTM wider than 256 per direction (BASELINE configs[3]: Whisper-small + full head, H = 384;
    configs[2]: WavLM-large + BiLSTM, H = 512): the three-pass recurrence in its four-wave form (csrc/lstm.hip, SELFGX) -- round 3 kept
    the bf16 recurrence there and said so only in a comment."""
import numpy as np
import pytest
import torch

from oracle import wfl_oracle as O
import synthetic as synth
from test_gpu_model import _build, _note, _oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("idx,B,L", [(3, 2, 160000), (2, 2, 48000)])
def test_precision_high_with_a_wide_bilstm(idx, B, L):
    cfg = synth.baseline_config(idx)
    cfg["model"]["precision"] = "high"
    m, labels, sd_np = _build(cfg, 70, seed=40 + idx)
    assert m.effective_precision().startswith("high (every product")
    wav = synth.make_batch(900 + idx, B, L, seed=40 + idx)
    lang = (np.arange(B) % 2).astype(np.int64)
    out = m.label(torch.from_numpy(wav).cuda(), lang, threshold=0.5, want_logits=True, want_hidden=True)
    m.check(B, L)
    lg, of, hid = _oracle(cfg, labels, sd_np, wav, lang)
    std = float(lg.std())
    err = (out.logits.cpu() - lg).abs()
    ids_ref, maxp_ref, arg_ref, margin = O.tags_from_logits(lg, labels.index("O"), 0.5)
    mism = out.argmax.cpu().long() != arg_ref
    _note(f"precision_high_wide_bilstm_cfg{idx + 1}", logit_std=std, logits_max=err.max(), logits_mean=err.mean(),
          offsets_max=(out.offsets.cpu() - of).abs().max(), raw_mismatches=int(mism.sum()), frames=int(mism.numel()))
    # the default build of these models sits at 0.30 / 0.067 (cfg4) and 0.024 / 0.005 (cfg3, logits ~ 20 x smaller); three passes
    # everywhere, the recurrence included, must land where the narrow models do: within 1e-3 of the logit scale
    scale = max(1.0, std / 6.5) if std > 1.0 else std / 6.5
    assert err.max() <= 2e-3 * max(scale, 0.05) and err.mean() <= 4e-4 * max(scale, 0.05), (float(err.max()), float(err.mean()), std)
    assert (out.offsets.cpu() - of).abs().max() <= 2e-4
    assert int(mism[margin > 0.005 * std / 6.5].sum()) == 0
    assert int(mism.sum()) <= 1
    # batch invariance and determinism of the new kernel form
    one = m.label(torch.from_numpy(wav[1:2]).cuda(), lang[1:2], threshold=0.5, want_logits=True)
    assert torch.equal(one.logits[0], out.logits[1])


# ------------------------------------------------------------------------------------------------------------------------------------
# csrc/gemm_mx.hip and its operand producer (csrc/norm.hip rows_fp8_kernel) at the operator level, through the C ABI
# (wfl_op_gemm_mx / wfl_op_rows_fp8): the block-scaled fp8 MFMA against a float64 product of the very bytes it is given.
# ------------------------------------------------------------------------------------------------------------------------------------
import ctypes as C  # noqa: E402

import torch.nn.functional as F  # noqa: E402

import gpu_util as G  # noqa: E402
from wfl_asr_amd import _lib  # noqa: E402


def _e4m3(x):
    return x.clamp(-448, 448).to(torch.float8_e4m3fn)


def _rows_fp8(x_rows, gamma=None, beta=None, pair=False):
    """bf16 frame rows (G.Rows) -> (hi bytes [R, C], lo bytes or None, scale [R]) through wfl_op_rows_fp8."""
    R, Cn = x_rows.buf.shape
    hi = torch.zeros(R, Cn, dtype=torch.uint8, device="cuda")
    lo = torch.zeros(R, Cn, dtype=torch.uint8, device="cuda") if pair else None
    sc = torch.zeros(R, dtype=torch.float32, device="cuda")
    rc = G.lib().wfl_op_rows_fp8(G.ptr(x_rows.buf), Cn, None, G.ptr(gamma), G.ptr(beta), 1e-5, x_rows.lead, x_rows.B, x_rows.P, x_rows.T, Cn,
                                 G.ptr(hi), G.ptr(lo), Cn, G.ptr(sc), G.stream())
    _lib.check(rc, "wfl_op_rows_fp8")
    return hi, lo, sc


@pytest.mark.parametrize("norm", [False, True])
def test_rows_fp8_pair_carries_eight_significant_bits(norm):
    B, T, Cn = 2, 77, 1280
    g = torch.Generator().manual_seed(5)
    x0 = (torch.randn(B, T, Cn, generator=g) * torch.logspace(-2, 1.5, Cn)).cuda()      # channels over 3.5 decades
    x = G.Rows(B, T, Cn).set(x0)
    gamma = (1 + 0.1 * torch.randn(Cn, generator=g)).cuda() if norm else None
    beta = (0.1 * torch.randn(Cn, generator=g)).cuda() if norm else None
    hi, lo, sc = _rows_fp8(x, gamma, beta, pair=True)
    hi1, _, sc1 = _rows_fp8(x, gamma, beta, pair=False)
    torch.cuda.synchronize()
    assert torch.equal(hi, hi1) and torch.equal(sc, sc1)                  # the hi plane and the scale do not depend on the pair
    ref = x.get()                                                          # the bf16 values the kernel read
    if norm:
        ref = F.layer_norm(ref, (Cn,), gamma, beta, 1e-5)
    rows = (x.lead + torch.arange(B)[:, None] * x.P + torch.arange(T)[None, :]).reshape(-1).cuda()
    s = sc[rows][:, None]
    h = hi[rows].view(torch.float8_e4m3fn).float()
    l = lo[rows].view(torch.float8_e4m3fn).float()
    ref = ref.reshape(-1, Cn)
    mx = ref.abs().amax(1, keepdim=True)
    assert torch.allclose(s, mx / 448.0, rtol=1e-5)                        # row maximum -> 448
    e1 = (h * s - ref).abs() / mx                                          # one e4m3 value: 2^-4 relative of the element, at worst
    e2 = ((h + l / 16) * s - ref).abs() / mx                               # the pair
    assert float(e1.max()) <= 2 ** -4 and float(e2.max()) <= 2 ** -8 + 1e-6, (float(e1.max()), float(e2.max()))
    big = ref.abs() > mx * 2 ** -5                                         # (elements in e4m3's normal range at this scale)
    rel2 = (((h + l / 16) * s - ref).abs() / ref.abs())[big]
    assert float(rel2.max()) <= 2 ** -8, float(rel2.max())                 # eight significant bits: what a bf16 operand carries


@pytest.mark.parametrize("pair", [False, True])
@pytest.mark.parametrize("mode", ["plain", "residual", "gelu_e4m3"])
def test_gemm_mx_against_a_float64_product_of_its_own_bytes(pair, mode):
    B, T, K, N = 3, 410, 1280, 768                       # 1290 rows: nine 160-row / seven 192-row tiles x 3 column tiles, tiles straddle clips
    g = torch.Generator().manual_seed(11 + pair)
    x = G.Rows(B, T, K).set((torch.randn(B, T, K, generator=g) * 0.7).cuda())
    hi, lo, sc = _rows_fp8(x, pair=pair)
    w8 = _e4m3(torch.randn(N, K, generator=g).cuda() * 60).view(torch.uint8).contiguous()
    ws = (torch.rand(N, generator=g).cuda() + 0.5) * 1e-3
    bias = (torch.randn(N, generator=g) * 0.2).cuda()
    out = G.Rows(B, T, N)
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    M = B * x.P
    res = res_lo = c_lo = c8 = c8lo = None
    act, inv = 0, 1.0
    if mode == "residual":
        r0 = torch.randn(B, T, N, generator=g).cuda()
        res = G.Rows(B, T, N).set(r0)
        res_lo = G.Rows(B, T, N).set((r0 - res.get()))                 # what the bf16 rounding of r0 left behind
        c_lo = G.Rows(B, T, N)
    if mode == "gelu_e4m3":
        act, inv = 1, 4.0
        c8 = torch.zeros(out.R, N, dtype=torch.uint8, device="cuda")
        c8lo = torch.zeros(out.R, N, dtype=torch.uint8, device="cuda") if pair else None
    rc = G.lib().wfl_op_gemm_mx(G.ptr(hi, x.lead * K), G.ptr(lo, x.lead * K) if pair else None, K, G.ptr(w8), G.ptr(ws), G.ptr(sc, x.lead), 1.0, M, N, K,
                                x.P, T, G.ptr(out.buf), N, out.lead, out.P, G.ptr(bias), G.ptr(res.buf) if res else None,
                                G.ptr(res_lo.buf) if res_lo else None, G.ptr(c_lo.buf) if c_lo else None, 1.0, act,
                                G.ptr(c8), G.ptr(c8lo), N, float(inv), G.ptr(status), G.stream())
    _lib.check(rc, "wfl_op_gemm_mx")
    torch.cuda.synchronize()
    assert int(status.item()) == 0
    rows = (x.lead + torch.arange(B)[:, None] * x.P + torch.arange(T)[None, :]).reshape(-1).cuda()
    a = hi[rows].view(torch.float8_e4m3fn).double()
    if pair:
        a = a + lo[rows].view(torch.float8_e4m3fn).double() / 16
    wd = w8.view(torch.float8_e4m3fn).double() * ws[:, None].double()
    ref = (a * sc[rows][:, None].double()) @ wd.T + bias.double()
    # The yardstick for the accumulation is the absolute-value product: the instruction adds its 128 products like an fp32 chain does
    # (measured with the un-rounded hi + lo output below: 4e-6 of sum_k |a_k w_k| at K = 1280 -- not the 12-13 bits some fp8 matrix units keep).
    mag = (a.abs() * sc[rows][:, None].double()) @ wd.abs().T
    if mode == "gelu_e4m3":
        ref = F.gelu(ref)
        orow = (out.lead + torch.arange(B)[:, None] * out.P + torch.arange(T)[None, :]).reshape(-1).cuda()
        got = c8[orow].view(torch.float8_e4m3fn).double()
        if pair:
            got = got + c8lo[orow].view(torch.float8_e4m3fn).double() / 16
        got = got / inv
        tol = (2 ** -8 if pair else 2 ** -4) * ref.abs().clamp_min(2 ** -6 / inv) + 1e-3 + 2e-5 * mag
        assert bool(((got - ref).abs() <= tol).all()), float(((got - ref).abs() - tol).max())
        assert bool((c8[:out.lead] == 0).all())                               # rows that are not frames are never written
        return
    if mode == "residual":
        ref = ref + res.get().reshape(-1, N).double() + res_lo.get().reshape(-1, N).double()
        got = out.get().reshape(-1, N).double() + c_lo.get().reshape(-1, N).double()
        err = (got - ref).abs()
        _note(f"gemm_mx_op_residual_pair{int(pair)}", err_over_mag_max=(err / mag).max(), err_max=err.max())
        assert bool((err <= 2 ** -15 * ref.abs() + 2e-5 * mag).all()), float((err / mag).max())     # hi + lo: sixteen significant bits of the sum
    else:
        got = out.get().reshape(-1, N).double()
        err = (got - ref).abs()
        _note(f"gemm_mx_op_plain_pair{int(pair)}", err_over_mag_max=(err / mag).max(), err_max=err.max())
        assert bool((err <= 2 ** -8 * ref.abs() + 2e-5 * mag).all()), float((err / mag).max())      # one bf16 rounding + the accumulation
    assert out.halo_is_zero()


def test_gemm_mx_reports_an_e4m3_output_that_does_not_fit():
    B, T, K, N = 1, 200, 512, 256
    x = G.Rows(B, T, K).set(torch.ones(B, T, K).cuda())
    hi, lo, sc = _rows_fp8(x)
    w8 = _e4m3(torch.ones(N, K).cuda()).view(torch.uint8).contiguous()
    ws = torch.ones(N, device="cuda")
    out = G.Rows(B, T, N)
    c8 = torch.zeros(out.R, N, dtype=torch.uint8, device="cuda")
    status = torch.zeros(1, dtype=torch.int32, device="cuda")
    for inv, want in ((0.5, 0), (1.0, 2)):                                  # every output is 512: 256 fits e4m3, 512 does not
        status.zero_()
        rc = G.lib().wfl_op_gemm_mx(G.ptr(hi, x.lead * K), None, K, G.ptr(w8), G.ptr(ws), G.ptr(sc, x.lead), 1.0, B * x.P, N, K, x.P, T, G.ptr(out.buf), N,
                                    out.lead, out.P, None, None, None, None, 1.0, 1, G.ptr(c8), None, N, float(inv), G.ptr(status), G.stream())
        _lib.check(rc, "wfl_op_gemm_mx")
        torch.cuda.synchronize()
        assert int(status.item()) == want, (inv, int(status.item()))


# ------------------------------------------------------------------------------------------------------------------------------------
# wfl_op_gemm_split: the exact-label Linear / Conv1d (operands and results as bf16 pairs, three products in one launch) at the operator
# level against a float64 product of the very pairs it is given.  The big shapes go to gemm256.hip's slice-by-slice walk (template TRI:
# every K slice's four tiles staged once for its three products) or, for the dense multi-tap conv, to the streaming kernel's
# tap-stationary mode.
# ------------------------------------------------------------------------------------------------------------------------------------
def _split(x):
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return hi, lo


class _PairRows:
    """hi and lo frame rows inside ONE allocation (the product keeps the low halves in a twin of the workspace, lo_delta away)."""

    def __init__(self, B, T, Cn):
        self.hi = G.Rows(B, T, Cn)
        both = torch.zeros(2 * self.hi.R, Cn, dtype=torch.bfloat16, device="cuda")
        self.hi.buf = both[:self.hi.R]
        self.lo = G.Rows(B, T, Cn)
        self.lo.buf = both[self.hi.R:]
        self.both = both

    def set(self, x):
        h, l = _split(x)
        self.hi.set(h.float())
        self.lo.set(l.float())
        return self

    def value(self):
        return self.hi.get().double() + self.lo.get().double()


@pytest.mark.parametrize("mode,B,T,K,N,taps", [
    ("plain", 3, 1500, 512, 512, 1),            # 4500 rows: TRI, 192- or 256-row tiles straddling clips
    ("gelu", 2, 1500, 512, 2048, 1),
    ("gelu", 16, 1500, 512, 2048, 1),           # cfg2's fc1 at full size: the 256-row tile (a 160 KiB ring)
    ("residual", 3, 1111, 2048, 512, 1),        # a ragged last row tile
    ("relu", 2, 1300, 768, 256, 3),             # three taps, dilation 2 (not the dense conv mode: taps two rows apart)
    ("glu", 2, 1500, 512, 1024, 1),
    ("residual", 1, 700, 512, 512, 1),          # one short clip: the same walk (chosen by shape, never by batch size)
])
def test_gemm_split_against_a_float64_product_of_its_own_pairs(mode, B, T, K, N, taps):
    g = torch.Generator().manual_seed(3 + K + N)
    cin = K // taps
    dil = 2 if taps > 1 else 0
    x = _PairRows(B, T, cin).set((torch.randn(B, T, cin, generator=g) * torch.logspace(-1, 1, cin)).cuda())
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    w_hi, w_lo = _split(w)
    w3 = torch.cat([w_hi, w_hi, w_lo], 1).contiguous()
    bias = torch.randn(N, generator=g).cuda()
    n_out = N // 2 if mode == "glu" else N
    out = _PairRows(B, T, n_out)
    res = _PairRows(B, T, n_out).set(torch.randn(B, T, n_out, generator=g).cuda() * 4) if mode == "residual" else None
    act = {"gelu": 1, "relu": 2}.get(mode, 0)
    P = x.hi.P
    lead = x.hi.lead - (taps // 2) * dil            # tap j of output frame t reads row t + (j - taps // 2) * dil (zero halos)
    rc = G.lib().wfl_op_gemm_split(G.ptr(x.hi.buf, lead * cin), G.ptr(x.lo.buf, lead * cin), cin, cin if taps > 1 else 0, dil * cin, G.ptr(w3),
                                   B * P, N, K, N, P, T, G.ptr(out.hi.buf), G.ptr(out.lo.buf), n_out, out.hi.lead, out.hi.P, G.ptr(bias),
                                   G.ptr(res.hi.buf) if res else None, G.ptr(res.lo.buf) if res else None, n_out, 1.0, act, int(mode == "glu"),
                                   G.stream())
    _lib.check(rc, "wfl_op_gemm_split")
    torch.cuda.synchronize()
    xh, xl = x.hi.get().double(), x.lo.get().double()
    if taps > 1:
        def unfold(v):
            v = F.pad(v, (0, 0, (taps // 2) * dil, (taps // 2) * dil))
            return torch.cat([v[:, j * dil:j * dil + T] for j in range(taps)], 2)
        xh, xl = unfold(xh), unfold(xl)
    wh, wl = w_hi.double(), w_lo.double()
    ref = xh @ wh.T + xl @ wh.T + xh @ wl.T + bias.double()           # the three products the mode computes (lo x lo is below 2^-16)
    mag = (xh.abs() + xl.abs()) @ (wh.abs() + wl.abs()).T + bias.abs().double()
    if mode == "glu":
        r = ref.reshape(B, T, N // 32, 2, 16)
        ref = (r[..., 0, :] * torch.sigmoid(r[..., 1, :])).reshape(B, T, n_out)
        mag = mag.reshape(B, T, N // 32, 2, 16)[..., 0, :].reshape(B, T, n_out)
    elif mode == "gelu":
        ref = F.gelu(ref)
    elif mode == "relu":
        ref = F.relu(ref)
    elif mode == "residual":
        ref = ref + res.value()
    got = out.value()
    err = (got - ref).abs()
    _note(f"gemm_split_{mode}_K{K}_N{N}_taps{taps}_B{B}", err_over_mag_max=(err / mag).max(), err_max=err.max())
    # sixteen significant bits of the result + an fp32 accumulation over K (measured ~1e-6 of sum |a||w|)
    assert bool((err <= 2 ** -15 * ref.abs() + 4e-6 * mag + 1e-30).all()), float((err / mag).max())
    assert out.hi.halo_is_zero() and out.lo.halo_is_zero()
