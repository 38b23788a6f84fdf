"""The arithmetic coder's chain divides by multiplying (gz_host.cpp builds the table, gz_kernels_arith.h:d_chain_step uses it):
    range / tot == mulhi (magic, range + inc) >> shift          with a 32-bit magic per divisor.
This restates the table construction in numpy and checks the identity for every divisor the model total can take, on the
numerators where a reciprocal scheme breaks first (multiples of the divisor and their neighbours, the top of the
32-bit range). The product's own table is exercised end to end by the parity tests."""
import numpy as np


def build(dmax):
    d = np.arange(1, dmax, dtype=np.uint64)
    L = np.floor(np.log2(d.astype(np.float64))).astype(np.uint64)
    L = np.where((np.uint64(1) << L) > d, L - np.uint64(1), L)
    L = np.where((np.uint64(1) << (L + np.uint64(1))) <= d, L + np.uint64(1), L)
    pow2 = (d & (d - np.uint64(1))) == 0
    num = np.uint64(1) << (np.uint64(32) + L)
    md, e = num // d, num % d
    up = (d - e) <= (np.uint64(1) << L)
    magic = np.where(pow2, np.uint64(0xffffffff), np.where(up, md + np.uint64(1), md))
    inc = np.where(pow2, np.uint64(1), np.where(up, np.uint64(0), np.uint64(1)))
    return d, magic, L, inc


def test_reciprocal_is_exact_for_every_model_total():
    d, magic, shift, inc = build(65536 + 32)
    assert magic.max() <= 0xffffffff
    top = np.uint64(0xffffffff)
    for k in range(0, 400):
        q = top // d - np.uint64(k % 200)                       # multiples of d near the top ...
        for n in (q * d, q * d - np.uint64(1), q * d + (d - np.uint64(1)), top - np.uint64(k + 1),
                  np.uint64(1 << 24) + np.uint64(k) * d, (np.uint64(k) * np.uint64(2654435761)) & top):
            n = np.minimum(n, top - inc)                        # (range + inc must not wrap: the chain treats range = 2^32-1 itself)
            got = ((magic * (n + inc)) >> np.uint64(32)) >> shift
            assert np.array_equal(got, n // d), k
