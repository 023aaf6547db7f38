#!/usr/bin/env python3
"""Fit of the GELU kernel form used by vtamiq_amd/csrc/dev_common.h gelu_erf: g(a) = 0.5 erfc(a / sqrt 2) = 2^(-1 - a Q(a)), Q a polynomial
fitted (Lawson-weighted Chebyshev least squares) to the minimax of the GELU absolute error, then checked with fp32 Horner evaluation."""
import numpy as np
from scipy.special import erfc, erf
from numpy.polynomial import chebyshev as Ch, polynomial as Po
# g(a) = 0.5*erfc(a/sqrt2) = 2^(-1 - a*Q(a)),  a in [0, A]
A = 5.7
def target(a):  # Q(a) = (-log2(0.5 erfc(a/√2)) - 1)/a
    return (-np.log2(0.5*erfc(a/np.sqrt(2))) - 1)/a
a = (np.cos(np.linspace(0, np.pi, 4001))*0.5+0.5)*A
a = a[a>1e-6]
for deg in (5,6,7,8,9):
    # weighted least squares in Chebyshev basis, then iterate weights (Lawson) to approx minimax on the GELU abs error
    t = 2*a/A-1
    w = np.ones_like(a)
    y = target(a)
    # error in gelu: d(gelu) = a*g*ln2*a*dQ -> weight = a^2 * g
    gw = a*a*0.5*erfc(a/np.sqrt(2))
    for it in range(60):
        c = Ch.chebfit(t, y, deg, w=w*gw)
        err = np.abs(Ch.chebval(t, c)-y)*gw
        w = w*(err/err.max()+1e-3); w/=w.max()
    p = Ch.cheb2poly(c)  # in t
    # convert to polynomial in a: t = 2a/A - 1
    pa = np.zeros(1)
    from numpy.polynomial import Polynomial as P
    poly_t = P(p); sub = P([-1, 2/A]); pa = poly_t(sub).coef
    # evaluate in float32 Horner and compare gelu
    x = np.linspace(-8, 8, 400001).astype(np.float32)
    ax = np.abs(x)
    axc = np.minimum(ax, np.float32(A))
    q = np.float32(pa[-1])*np.ones_like(axc)
    for k in range(len(pa)-2, -1, -1):
        q = (q*axc + np.float32(pa[k])).astype(np.float32)
    arg = (-(axc*q) - np.float32(1)).astype(np.float32)
    g = np.exp2(arg.astype(np.float64)).astype(np.float32)
    gel = (np.maximum(x,0) - ax*g).astype(np.float32)
    ref = 0.5*x.astype(np.float64)*(1+erf(x.astype(np.float64)/np.sqrt(2)))
    e = np.abs(gel-ref)
    print(deg, "max abs err", e.max(), "at x=", x[e.argmax()], " max rel to max(1,|x|)", (e/np.maximum(1,ax)).max(), "coeffs", [float(np.float32(v)) for v in pa])
