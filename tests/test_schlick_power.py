"""Schlick's (1 - cos)^5 (material.h:73 calls pow(x, 5)) on the device is pow5() of csrc/rtk_trace.hip: products carried
with their rounding errors, summed once.  This file restates that arithmetic (two_prod by Veltkamp / Dekker splitting stands in
for the fused multiply-add) and checks it against EXACT rational arithmetic -- it must be the correctly rounded power for every
argument -- and measures how often the reference's own pow (glibc, through the g++-built reference and oracle) is that value:
the only cases in which device and reference can disagree on the reflectance at all."""
import ctypes
import fractions
import random


def _split(a):
    c = 134217729.0 * a
    hi = c - (c - a)
    return hi, a - hi


def _two_prod(a, b):  # a * b = p + e exactly (what fma(a, b, -p) returns on the device)
    p = a * b
    ah, al = _split(a)
    bh, bl = _split(b)
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def pow5_device(x):  # csrc/rtk_trace.hip pow5(double), operation for operation
    x2, e2 = _two_prod(x, x)
    x4, e4 = _two_prod(x2, x2)
    e4 = e4 + 2.0 * (x2 * e2)
    x5, e5 = _two_prod(x4, x)
    e5 = e5 + e4 * x
    return x5 + e5


def _arguments(n, seed):
    rng = random.Random(seed)
    for k in range(n):
        if k % 8 == 0:
            yield 1.0 - rng.random() * 2.0 ** -rng.randint(1, 40)      # grazing incidence: 1 - cos close to 1
        elif k % 8 == 1:
            yield rng.random() * 2.0 ** -rng.randint(1, 60)            # head-on: 1 - cos tiny
        else:
            yield rng.random()


def test_device_power_is_the_correctly_rounded_fifth_power():
    for x in _arguments(60000, 1):
        exact = fractions.Fraction(x) ** 5
        assert pow5_device(x) == exact.numerator / exact.denominator, x     # int / int is correctly rounded in Python


def test_reference_pow_is_that_value_in_all_but_a_per_mille_of_cases():
    libm = ctypes.CDLL("libm.so.6")
    libm.pow.restype = ctypes.c_double
    libm.pow.argtypes = [ctypes.c_double, ctypes.c_double]
    n = differs = plain_differs = 0
    for x in _arguments(60000, 2):
        exact = fractions.Fraction(x) ** 5
        want = exact.numerator / exact.denominator
        got = libm.pow(x, 5.0)
        n += 1
        differs += got != want
        plain_differs += (x * x) * (x * x) * x != want
        assert abs(got - want) <= abs(want) * 2.3e-16     # never more than an ulp apart
    assert differs < 0.003 * n, (differs, n)              # measured: 0.08 %
    assert plain_differs > 0.3 * n                         # the three plain products of rounds 1-2: about half
