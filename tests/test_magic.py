"""The arithmetic coder's chain divides by multiplying, in double precision with truncation (gz_host.cpp builds the table,
gz_kernels_arith.h:d_chain_step and the loop of gz_chain_asm.h use it):
    range / tot == low word of fma (range * 2^-7, inv, 2^52) rounded toward zero,   inv = 2^7 / tot rounded UP to a double
The fma forms range * 2^-7 * inv exactly, adds 2^52 and truncates to the 53 bits of a double, i.e. to an integer: the result is
2^52 + floor (range * inv / 2^7) computed without error. This restates the table construction and checks the identity in exact
integer arithmetic for every divisor the model total can take, on the numerators where a reciprocal scheme breaks first (multiples
of the divisor and the numbers just below them, up to the top of the 32-bit range). The product's own table and kernels are
exercised end to end by the parity tests; the renormalisation by exponent bits is checked here as well."""
import math
import random
import struct
from fractions import Fraction


def inv_of(d):
    inv = 128.0 / d
    if Fraction(inv) * d < 128:                                  # (the host asks fma (inv, d, -128) for the sign)
        inv = math.nextafter(inv, math.inf)
    return inv


def quotient(n, inv):
    m, e = math.frexp(inv)                                       # inv = m * 2^e exactly, m * 2^53 an integer
    M = int(m * (1 << 53))
    s = 53 - e + 7                                               # n * 2^-7 * inv = n * M / 2^s
    return (n * M) >> s if s >= 0 else (n * M) << -s


def test_reciprocal_is_exact_for_every_model_total():
    rnd = random.Random(1)
    top = 0xffffffff
    for d in range(1, 65536 + 32):
        inv = inv_of(d)
        assert Fraction(inv) * d >= 128 and Fraction(math.nextafter(inv, 0.0)) * d < 128 or Fraction(inv) * d == 128
        q = top // d
        ns = [top, q * d, q * d - 1, (q - 1) * d, (q - 1) * d + d - 1, 1 << 24, ((1 << 24) // d + 1) * d - 1, ((1 << 24) // d + 1) * d]
        for _ in range(12):
            k = rnd.randrange((1 << 24) // d + 1, q + 1)
            ns += [k * d, k * d - 1, min(top, k * d + rnd.randrange(d))]
        for n in ns:
            assert quotient(n, inv) == n // d, (d, n)


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
