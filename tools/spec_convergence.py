#!/usr/bin/env python3
"""CPU experiment behind DESIGN.md section 6 (0b): does the adaptive model (c_simple_model.h: +16 per symbol, all frequencies halved when
the total passes 65519, one bubble step to the left) FORGET its starting state? A model is run over a stream from the start ("true"), a second
one from a flat start W symbols before a check point; at the check point the two are compared: frequencies equal? list order equal? would
the NEXT 100 000 symbols get the same (cum, freq, total) triples? Usage: python tools/spec_convergence.py"""
import sys
import numpy as np

STEP, LIMIT = 16, 65519


class Model:
    def __init__(self, nsym):
        self.sym = list(range(nsym)); self.f = [1] * nsym; self.tot = nsym

    def code(self, s):
        i = self.sym.index(s)
        cum = sum(self.f[:i]); out = (cum, self.f[i], self.tot)
        self.f[i] += STEP; self.tot += STEP
        if self.tot > LIMIT:
            self.f = [x - (x >> 1) for x in self.f]; self.tot = sum(self.f)
        if i and self.f[i] > self.f[i - 1]:
            self.f[i], self.f[i - 1] = self.f[i - 1], self.f[i]; self.sym[i], self.sym[i - 1] = self.sym[i - 1], self.sym[i]
        return out


def stream(kind, n, seed):
    r = np.random.default_rng(seed)
    if kind == "qual40":      # 40 levels, geometric-ish (a quality context)
        p = 0.85 ** np.arange(40); p /= p.sum(); return r.choice(40, n, p=p)
    if kind == "bin4":        # 4 levels + 4 rare ones
        p = np.array([0.6, 0.25, 0.1, 0.0499, 1e-4 / 4, 1e-4 / 4, 1e-4 / 4, 1e-4 / 4]); p /= p.sum(); return r.choice(8, n, p=p)
    if kind == "uniform200":  # near-uniform wide alphabet
        return r.integers(0, 200, n)
    if kind == "zipf256":     # wide alphabet with a long dormant tail
        p = 1.0 / (1 + np.arange(256)) ** 1.5; p /= p.sum(); return r.choice(256, n, p=p)
    raise ValueError(kind)


def main():
    N, TAIL = 600000, 100000
    for kind in ("qual40", "bin4", "uniform200", "zipf256"):
        d = stream(kind, N + TAIL, 1).tolist()
        nsym = max(d) + 1
        for W in (20000, 100000, 300000):
            res = []
            for cp in (300000, 400000, 500000, 600000):
                t = Model(nsym)
                for s in d[:cp]: t.code(s)
                sp = Model(nsym)
                for s in d[cp - W:cp]: sp.code(s)
                same_f = sorted(zip(t.sym, t.f)) == sorted(zip(sp.sym, sp.f)) and t.tot == sp.tot
                same_order = t.sym == sp.sym
                same_out = all(t.code(s) == sp.code(s) for s in d[cp:cp + TAIL])
                res.append((same_f, same_order, same_out))
            print("%-11s W %6d: frequencies equal %d/4, order equal %d/4, next %d triples equal %d/4" %
                  (kind, W, sum(r[0] for r in res), sum(r[1] for r in res), TAIL, sum(r[2] for r in res)), flush=True)


if __name__ == "__main__":
    main()
