"""Coefficients of fast_exp (csrc/device_common.hpp): degree-n interpolant of e^r at the
Chebyshev nodes of |r| <= ln2 / 2 in 80-digit arithmetic, and the maximum relative error of
the double-rounded coefficients (degree 11: 1.7e-17; Taylor to r^13: 5.2e-18).

    python profiles/micro/exp_poly.py
"""
# degree-n polynomial for exp on [-a, a], a = ln2/2, by Chebyshev interpolation in 80-digit
# arithmetic; report the max relative error of the double-rounded coefficients
from decimal import Decimal as Dm, getcontext
import math
getcontext().prec = 80
LN2 = Dm(2).ln()
a = LN2 / 2
def dexp(x): return x.exp()
def cheb_interp_monomial(n):
    # nodes x_k = a cos(pi (k + 1/2)/(n+1)), solve Vandermonde via Newton divided differences
    PI = Dm('3.14159265358979323846264338327950288419716939937510582097494459230781640628620899')
    def dcos(x):
        # Taylor
        s, term, k = Dm(0), Dm(1), 0
        x2 = x * x
        while abs(term) > Dm(10) ** -75:
            s += term
            k += 2
            term = -term * x2 / (k * (k - 1))
        return s
    xs = [a * dcos(PI * (Dm(k) + Dm('0.5')) / (n + 1)) for k in range(n + 1)]
    ys = [dexp(x) for x in xs]
    # divided differences
    coef = ys[:]
    for j in range(1, n + 1):
        for i in range(n, j - 1, -1):
            coef[i] = (coef[i] - coef[i - 1]) / (xs[i] - xs[i - j])
    # convert Newton form to monomial
    poly = [Dm(0)] * (n + 1)
    poly[0] = coef[n]
    deg = 0
    for k in range(n - 1, -1, -1):
        # poly = poly * (x - xs[k]) + coef[k]
        new = [Dm(0)] * (n + 1)
        for d in range(deg + 1):
            new[d + 1] += poly[d]
            new[d] -= poly[d] * xs[k]
        new[0] += coef[k]
        poly = new
        deg += 1
    return poly
def max_err(cs_double, n):
    worst = Dm(0)
    N = 4001
    for i in range(N):
        x = -a + 2 * a * Dm(i) / (N - 1)
        xd = Dm(float(x))
        p = Dm(0)
        for c in reversed(cs_double):
            p = p * xd + Dm(c)
        e = abs(p / dexp(xd) - 1)
        worst = max(worst, e)
    return worst
for n in (10, 11, 12):
    poly = cheb_interp_monomial(n)
    cs = [float(c) for c in poly]
    print(n, 'max rel err (exact arithmetic, double coeffs): %.3e' % max_err(cs, n))
    cs2 = cs[:]; cs2[0] = 1.0; cs2[1] = 1.0
    print(n, '  with c0 = c1 = 1: %.3e' % max_err(cs2, n))
    if n == 11:
        print([c.hex() for c in cs]); print(['%.17g' % c for c in cs])
taylor = [1.0 / math.factorial(k) for k in range(14)]
print('taylor13 %.3e' % max_err(taylor, 13))
