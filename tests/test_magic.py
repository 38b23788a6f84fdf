"""The arithmetic coder's chain divides by multiplying, in double precision with truncation (gz_kernels_arith.h: d_record_inv makes the
reciprocal, d_chain_step and the loop of gz_chain_asm.h use it):
    range / tot == low word of fma (range * 2^-7, inv, 1.0) rounded toward zero        for ANY double inv in
    [2^-45 / tot, 2^-45 / tot * (1 + 2^-33)]
The fma forms range * 2^-7 * inv exactly, adds 1 and truncates to the 53 bits of a double, i.e. to a multiple of 2^-52: the result
is 1 + floor (range * inv * 2^45) * 2^-52 computed without error. The model kernel's inv is the hardware's reciprocal seed and one
Newton step with the constant 1 + 2^-34 (about 2^-34 above 1 / tot), its 16 low bits replaced by the symbol's cum: inside that
interval with room to spare (tests/test_gpu.py::test_record_reciprocals checks every total's value made on the device). This checks
the identity in exact integer arithmetic for every divisor the model total can take, at both ends of the interval, on the
numerators where a reciprocal scheme breaks first (multiples of the divisor and the numbers just below them, up to the top of the
32-bit range); the renormalisation by exponent bits is checked here as well."""
import math
import random
import struct
from fractions import Fraction


def interval_ends(d):
    """the smallest double >= 2^-45 / d and the largest one <= 2^-45 / d * (1 + 2^-33)"""
    exact = Fraction(1, d << 45)
    lo = float(exact)
    while Fraction(lo) < exact:
        lo = math.nextafter(lo, math.inf)
    top = exact * (1 + Fraction(1, 1 << 33))
    hi = float(top)
    while Fraction(hi) > top:
        hi = math.nextafter(hi, 0.0)
    return lo, hi


def quotient(n, inv):
    m, e = math.frexp(inv)                                       # inv = m * 2^e exactly, m * 2^53 an integer
    M = int(m * (1 << 53))
    s = 53 - e + 7 - 52                                          # n * 2^-7 * inv * 2^52 = n * M / 2^s
    return (n * M) >> s if s >= 0 else (n * M) << -s


def admissible(d, inv):
    return Fraction(1, d << 45) <= Fraction(inv) <= Fraction(1, d << 45) * (1 + Fraction(1, 1 << 33))


def test_reciprocal_is_exact_for_every_model_total():
    rnd = random.Random(1)
    top = 0xffffffff
    for d in range(1, 65536 + 32):
        for inv in interval_ends(d):
            assert admissible(d, inv)
            q = top // d
            ns = [top, q * d, q * d - 1, (q - 1) * d, (q - 1) * d + d - 1, 1 << 24, ((1 << 24) // d + 1) * d - 1, ((1 << 24) // d + 1) * d]
            for _ in range(6):
                k = rnd.randrange((1 << 24) // d + 1, q + 1)
                ns += [k * d, k * d - 1, min(top, k * d + rnd.randrange(d))]
            for n in ns:
                assert quotient(n, inv) == n // d, (d, n, inv)


def test_renormalisation_by_exponent_bits():
    """x * 2^-7 as a double with the high word's exponent forced to 0x410 | (its low 3 bits) == x shifted left by whole bytes until
    it is >= 2^24 (x >= 2^8), times 2^-7"""
    rnd = random.Random(2)
    for _ in range(200000):
        x = rnd.randrange(1 << rnd.randrange(9, 33))
        if x < 256:
            x += 256
        want = x
        while want < (1 << 24):
            want <<= 8
        bits = struct.unpack("<Q", struct.pack("<d", x / 128.0))[0]
        hi = ((bits >> 32) & 0x007fffff) | 0x41000000
        got = struct.unpack("<d", struct.pack("<Q", hi << 32 | (bits & 0xffffffff)))[0]
        assert got * 128.0 == want, x
