"""Go's PORTABLE math.Exp, restated (test infrastructure, like everything under oracle/).

The reference's softmax calls math.Exp (src/ml/operations_impl.go:498, :506) -- the Go standard library, a dependency that is not in /root/reference (Go 1.22
`math`).  Its portable implementation (src/math/exp.go, "The original C code, the long comment, and the constants below are from FreeBSD's /usr/src/lib/msun/src/e_exp.c")
is the published fdlibm algorithm: argument reduction x = k ln2 + r with a two-part ln2, a degree-5 Remez polynomial in r^2, reconstruction by Ldexp.  Restated here in
IEEE float64 operations (Python floats: no fused multiply-add) so that the distance between THREE faithful implementations can be measured over the 65536 bf16 inputs a softmax
can see: this one, the host libm the C oracle calls, and the device's ocml (lnb_op_exp_table).  On amd64 / arm64 / s390x Go replaces the portable code by assembly
(exp_amd64.s, exp_arm64.s: the same algorithm with fused multiply-adds on arm64), so not even the reference agrees with itself across platforms at the last ulp -- which is
the point: parity at this level is defined after the float32 narrowing of impl:506, where all of them coincide.

Unpinned in the strict sense of the task statement: no toolchain here can run Go, the constants are fdlibm's as published; what pins the restatement is that it is within one
ulp of the host libm on every one of the 65536 inputs (tests/test_exp_implementations.py) -- a mistyped constant would be off by thousands.
"""
import math

LN2_HI = 6.93147180369123816490e-01
LN2_LO = 1.90821492927058770002e-10
LOG2E = 1.44269504088896338700e+00
P1, P2, P3, P4, P5 = 1.66666666666666019037e-01, -2.77777777770155933842e-03, 6.61375632143793436117e-05, -1.65339022054652515390e-06, 4.13813679705723846039e-08
OVERFLOW, UNDERFLOW, NEAR_ZERO = 7.09782712893383973096e+02, -7.45133219101941108420e+02, 1.0 / (1 << 28)


def go_exp(x):
    """exp.go: func exp(x float64) float64 (special cases, reduction, expmulti)"""
    if x != x or x == math.inf:
        return x
    if x == -math.inf:
        return 0.0
    if x > OVERFLOW:
        return math.inf
    if x < UNDERFLOW:
        return 0.0
    if -NEAR_ZERO < x < NEAR_ZERO:
        return 1.0 + x
    k = int(LOG2E * x + 0.5) if x > 0 else int(LOG2E * x - 0.5) if x < 0 else 0
    hi = x - float(k) * LN2_HI
    lo = float(k) * LN2_LO
    r = hi - lo                                              # expmulti(hi, lo, k)
    t = r * r
    c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))))
    y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi)
    return math.ldexp(y, k)
