"""Thin helpers for the `-m gpu` parity tests: frame-row buffers (csrc/common.h layout) and ctypes calls
through the C ABI single-op entry points."""
import ctypes as C

import torch

from wfl_asr_amd import _lib


def lib():
    return _lib.load()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, elem_off=0):
    if t is None:
        return C.c_void_p(0)
    return C.c_void_p(t.data_ptr() + elem_off * t.element_size())


class Rows:
    """bf16 frame rows: row(b, t) = lead + b*P + t, zero halos."""

    def __init__(self, B, T, C_, halo=16, lead=16, tail=256, pitch=None, device="cuda"):
        self.B, self.T, self.C = B, T, C_
        self.P = pitch if pitch is not None else (T + halo + 7) // 8 * 8
        self.lead, self.tail = lead, tail
        self.R = lead + B * self.P + tail
        self.buf = torch.zeros(self.R, C_, dtype=torch.bfloat16, device=device)

    def set(self, x):                      # x [B, T, C] float
        v = self.buf[self.lead:self.lead + self.B * self.P].view(self.B, self.P, self.C)
        v[:, :self.T] = x.to(torch.bfloat16)
        return self

    def get(self):                         # -> [B, T, C] float32
        v = self.buf[self.lead:self.lead + self.B * self.P].view(self.B, self.P, self.C)
        return v[:, :self.T].float()

    def halo_is_zero(self):
        v = self.buf[self.lead:self.lead + self.B * self.P].view(self.B, self.P, self.C)
        return bool((v[:, self.T:] == 0).all() and (self.buf[:self.lead] == 0).all()
                    and (self.buf[self.lead + self.B * self.P:] == 0).all())


def pad_weight(w, bias=None):
    """[N, K] float -> bf16 [N128, K64] + fp32 bias [N128]."""
    N, K = w.shape
    Np, Kp = (N + 127) // 128 * 128, (K + 63) // 64 * 64
    wp = torch.zeros(Np, Kp, dtype=torch.bfloat16, device=w.device)
    wp[:N, :K] = w.to(torch.bfloat16)
    bp = torch.zeros(Np, dtype=torch.float32, device=w.device)
    if bias is not None:
        bp[:N] = bias
    return wp, bp


def gemm(A, a_off, lda, Wp, M, n_valid, P, T, Cout, c_ld, c_lead, c_pitch, bias=None, res=None, ldres=0, alpha=1.0,
         act=0, glu=0, out_f32=0, cin=0, tap_stride=0):
    N, K = Wp.shape
    rc = lib().wfl_op_gemm(ptr(A, a_off), lda, cin, tap_stride, ptr(Wp), M, N, K, n_valid, P, T, ptr(Cout), c_ld, c_lead,
                           c_pitch, ptr(bias), ptr(res), ldres, float(alpha), act, glu, out_f32, stream())
    _lib.check(rc, "wfl_op_gemm")


def gemm_ln(A, a_off, lda, Wp, M, n_valid, P, T, Cout, c_ld, c_lead, c_pitch, bias, ln_s, eps=1e-5, act=0):
    N, K = Wp.shape
    rc = lib().wfl_op_gemm_ln(ptr(A, a_off), lda, ptr(Wp), M, N, K, n_valid, P, T, ptr(Cout), c_ld, c_lead, c_pitch, ptr(bias),
                              ptr(ln_s), float(eps), act, stream())
    _lib.check(rc, "wfl_op_gemm_ln")


def attention(QK, ldqk, lead, V, v_off, ldv, O, ldo, B, T, P, heads, d):
    _lib.check(lib().wfl_op_attention(ptr(QK), ldqk, lead, ptr(V, v_off), ldv, ptr(O), ldo, B, T, P, heads, d, stream()), "wfl_op_attention")


def layernorm(x, y, g, b, eps, lead, B, P, T, Cn):
    _lib.check(lib().wfl_op_layernorm(ptr(x), Cn, ptr(y), Cn, ptr(g), ptr(b), float(eps), lead, B, P, T, Cn, stream()), "wfl_op_layernorm")


def tag_decide(logits, thr, o_id):
    rows, Cn = logits.shape
    ids = torch.empty(rows, dtype=torch.int32, device=logits.device)
    arg = torch.empty_like(ids)
    mp = torch.empty(rows, dtype=torch.float32, device=logits.device)
    _lib.check(lib().wfl_op_tag_decide(ptr(logits), Cn, rows, Cn, float(thr), o_id, ptr(ids), ptr(arg), ptr(mp), stream()), "wfl_op_tag_decide")
    return ids, arg, mp
