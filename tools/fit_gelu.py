"""Diagnostic (not product): minimax-style fit behind gelu_erf() in csrc/common.h (needs scipy)."""
import numpy as np
from scipy.special import erfc, erf
def gelu_ref(x): return 0.5*x*(1+erf(x/np.sqrt(2)))
# fit q(t) ~ log2(0.5*erfc(t/sqrt2)) with t=|x| directly (fold constants): h = |x| * exp2(q(|x|)); gelu = max(x,0) - h
t = np.linspace(0, 6.0, 6001)
y = np.log2(0.5*erfc(t/np.sqrt(2)))
w0 = np.maximum(t*erfc(t/np.sqrt(2)), 1e-7)
best=None
for deg in (5,6):
    w=w0.copy()
    for it in range(60):
        c = np.polyfit(t, y, deg, w=w)
        e = np.abs(np.polyval(c, t) - y) * w0
        w = w * (1 + 2*e/e.max()); w = w/w.max()*w0.max()
    c32 = c.astype(np.float32)
    x = np.linspace(-12, 12, 2400001).astype(np.float32)
    ax = np.abs(x)
    q = np.zeros_like(ax) + c32[0]
    for k in c32[1:]: q = (q*ax + k).astype(np.float32)
    h = (ax*np.exp2(q.astype(np.float64))).astype(np.float32)
    g = np.maximum(x,0) - h
    ref = gelu_ref(x.astype(np.float64))
    err = np.abs(g-ref)
    print(deg, "max abs err %.3g at %.3f" % (err.max(), x[err.argmax()]), "coeffs hi->lo:", ", ".join("%.9ef"%v for v in c32))
    # monotone decreasing check of q beyond fit range
    tt = np.linspace(6, 100, 1000); print("  q(6..100) max:", np.polyval(c, tt).max())
