import numpy as np, sys
STEP, LIMIT = 16, 65519
def sim(d, nsym, warm=20000):
    sym = list(range(nsym)); f = [1]*nsym; tot = nsym
    pos = {s:i for i,s in enumerate(sym)}
    ev=[]; ps=[]
    for s in d:
        i = pos[s]
        f[i] += STEP; tot += STEP
        if tot > LIMIT:
            f = [x - (x >> 1) for x in f]; tot = sum(f)
        e = 0
        if i and f[i] > f[i-1]:
            f[i], f[i-1] = f[i-1], f[i]; a, b = sym[i], sym[i-1]; sym[i], sym[i-1] = b, a; pos[a] = i-1; pos[b] = i; e = 1
        ev.append(e); ps.append(i)
    ev = ev[warm:]; ps = ps[warm:]
    nb = len(ev)//64; rounds = 0; events = 0
    for b in range(nb):
        E = ev[b*64:(b+1)*64]; P = ps[b*64:(b+1)*64]; events += sum(E)
        i0 = 0
        while i0 < 64:
            rounds += 1
            k = i0 + 1; evpos = set()
            if E[i0]: evpos.update((P[i0]-1, P[i0], P[i0]+1))
            while k < 64 and P[k] not in evpos:
                if E[k]: evpos.update((P[k]-1, P[k], P[k]+1))
                k += 1
            i0 = k
    return rounds/nb, events/nb
r = np.random.default_rng(1)
N = 120000
tests = {
 "uniform256": r.integers(0,256,N),
 "uniform200": r.integers(0,200,N),
 "uniform100": r.integers(0,100,N),
 "zipf256": None, "qual40": None, "geom90 (0.95)": None, "two-level 256 (16 hot 70%)": None,
}
p = 1.0/(1+np.arange(256))**1.5; p/=p.sum(); tests["zipf256"] = r.choice(256,N,p=p)
p = 0.85**np.arange(40); p/=p.sum(); tests["qual40"] = r.choice(40,N,p=p)
p = 0.95**np.arange(90); p/=p.sum(); tests["geom90 (0.95)"] = r.choice(90,N,p=p)
p = np.concatenate([np.full(16,0.7/16), np.full(240,0.3/240)]); tests["two-level 256 (16 hot 70%)"] = r.choice(256,N,p=p)
for k,d in tests.items():
    d = d.tolist(); rr, ee = sim(d, max(d)+1)
    print("%-28s events per batch %5.1f  rounds per batch %5.1f" % (k, ee, rr))
