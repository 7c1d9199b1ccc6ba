"""Study (not a test; CPU, numpy): what a frequency-domain form of the Conformer's dense k = 31 Conv1d (/root/reference/model.py:33) would
cost in accuracy.  Overlap-save with 64-point blocks (34 new frames each): forward / inverse DFT in fp32, the spectra of the activations
AND of the weights rounded to bf16 in front of the per-bin complex GEMM (the only place bf16 MFMAs would be used), fp32 accumulation.
Yardstick: the error the default build already carries in this layer -- operands rounded to bf16 in the time domain.
usage: python tests/study_fftconv.py"""
import numpy as np


def bf16(x):
    x = np.asarray(x, np.float32)
    u = x.view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def main():
    rng = np.random.RandomState(0)
    C, K, T, NB = 256, 31, 600, 64                       # (a quarter of the real layer's channels: the error statistics do not depend on it)
    L = NB - K + 1
    x = rng.randn(T, C).astype(np.float32) * np.exp(rng.randn(C) * 0.5).astype(np.float32)      # channels of unequal scale
    w = (rng.randn(C, C, K) / np.sqrt(C * K)).astype(np.float32)                                   # [out][in][tap]
    pad = K // 2

    def direct(xx, ww):
        xp = np.pad(xx.astype(np.float64), ((pad, pad), (0, 0)))
        y = np.zeros((T, C))
        for k in range(K):
            y += xp[k:k + T] @ ww[:, :, k].astype(np.float64).T
        return y

    y_ref = direct(x, w)                                 # fp32 operands, float64 sums: the reference's arithmetic
    y_bf = direct(bf16(x), bf16(w))                      # the default build: operands rounded to bf16, exact products
    # overlap-save: block b covers input frames [b L - pad, b L - pad + NB), yields outputs [b L, b L + L)
    xb = bf16(x)
    xp = np.pad(xb, ((pad, pad + NB), (0, 0)))
    nblk = (T + L - 1) // L
    blocks = np.stack([xp[b * L:b * L + NB] for b in range(nblk)])                  # [blk][NB][C]
    X = np.fft.rfft(blocks.astype(np.float32), axis=1).astype(np.complex64)          # fp32 DFT of the bf16 activations
    wk = np.zeros((C, C, NB), np.float64)
    wk[:, :, :K] = w[:, :, ::-1]                          # correlation -> convolution with the flipped taps
    Wf = np.fft.rfft(wk, axis=2)                          # weights: transformed once at load time, in float64

    def run(Xr, Xi, Wr, Wi):
        Yr = np.einsum("bfc,ocf->bfo", Xr, Wr) - np.einsum("bfc,ocf->bfo", Xi, Wi)
        Yi = np.einsum("bfc,ocf->bfo", Xr, Wi) + np.einsum("bfc,ocf->bfo", Xi, Wr)
        yb = np.fft.irfft(Yr + 1j * Yi, n=NB, axis=1)
        return np.concatenate([yb[b, K - 1:K - 1 + L] for b in range(nblk)])[:T]

    f64 = lambda a: a.astype(np.float64)
    y_exact = run(f64(X.real), f64(X.imag), Wf.real, Wf.imag)                       # sanity: the transform itself
    y_fft = run(f64(bf16(X.real)), f64(bf16(X.imag)), f64(bf16(Wf.real)), f64(bf16(Wf.imag)))
    # weights as pairs (two passes on the weight side), spectra of the activations single
    wr_lo, wi_lo = bf16(Wf.real - bf16(Wf.real)), bf16(Wf.imag - bf16(Wf.imag))
    y_fft_wpair = run(f64(bf16(X.real)), f64(bf16(X.imag)), f64(bf16(Wf.real)) + f64(wr_lo), f64(bf16(Wf.imag)) + f64(wi_lo))
    s = y_ref.std()
    rel = lambda a, b: (np.abs(a - b).max() / s, np.sqrt(((a - b) ** 2).mean()) / s)
    print("output std %.3f" % s)
    print("transform alone (bf16 activations, exact spectra, fp32 weights) vs the default : max %.2e  rms %.2e   (= the weights' rounding)" % rel(y_exact, y_bf))
    print("default build (bf16 operands, time domain)  vs the reference               : max %.2e  rms %.2e" % rel(y_bf, y_ref))
    print("frequency domain, bf16 spectra both sides   vs the reference               : max %.2e  rms %.2e" % rel(y_fft, y_ref))
    print("frequency domain, bf16 spectra both sides   vs the default build           : max %.2e  rms %.2e" % rel(y_fft, y_bf))
    print("frequency domain, weight spectra as pairs   vs the reference               : max %.2e  rms %.2e" % rel(y_fft_wpair, y_ref))


if __name__ == "__main__":
    main()
