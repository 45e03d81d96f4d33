import random
LB, NL = 29, 9
M = (1 << LB) - 1
R = 1 << (LB * NL)
def limbs(x): return [(x >> (LB * i)) & M for i in range(NL)]
def val(l): return sum(v << (LB * i) for i, v in enumerate(l))
for p in (21888242871839275222246405745257275088548364400416034343698204186575808495617,
          52435875175126190479447740508185965837690552500527637822603658699938581184513):
    NP = (-pow(p, -1, R)) % R
    pl = limbs(p)
    rnd = random.Random(1)
    worst = 0
    for it in range(20000):
        # x: loose / uncarried limbs < 2^31, value < 64p
        xv = rnd.randrange(64 * p)
        xl = limbs(xv)
        # perturb into non-normalised form with limbs up to 2^31 keeping the value: move carries down
        for k in range(NL - 1, 0, -1):
            if xl[k] > 0 and rnd.random() < 0.5:
                t = rnd.randrange(1, min(xl[k], 3) + 1)
                if xl[k - 1] + (t << LB) < (1 << 31):
                    xl[k] -= t; xl[k - 1] += t << LB
        assert val(xl) == xv and max(xl) < 1 << 31
        wv = rnd.randrange(p)
        wl = limbs(wv)
        wpp = limbs(wv * NP % R)
        # step 1: m = low9(x * w'')
        carry = 0; m = []
        for k in range(NL):
            col = sum(xl[i] * wpp[k - i] for i in range(k + 1))
            assert col < 1 << 64
            t = col + carry
            assert t < 1 << 64
            m.append(t & M); carry = t >> LB
        # step 2: columns 7..16 of x*w + m*p
        T = {}
        for k in range(7, 17):
            T[k] = sum(xl[i] * wl[k - i] + m[i] * pl[k - i] for i in range(NL) if 0 <= k - i < NL)
            assert T[k] < 1 << 64
        u = T[8] + (T[7] >> LB)
        assert u + (1 << 28) < 1 << 64
        Z = (u + (1 << 28)) >> LB
        r = []
        t = Z
        for k in range(9, 17):
            t += T[k]
            assert t < 1 << 64
            r.append(t & M); t >>= LB
        r.append(t)
        rv = val(r)
        want = (xv * wv + val(m) * p) // R
        assert (xv * wv + val(m) * p) % R == 0
        assert rv == want, (it, rv, want)
        assert rv % p == xv * wv * pow(R, -1, p) % p
        assert rv < 2 * p and r[8] < 1 << LB
        worst = max(worst, rv / p)
    print("ok", hex(NP)[:12], worst)
