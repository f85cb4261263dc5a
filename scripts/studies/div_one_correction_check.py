#!/usr/bin/env python3
"""Is ONE residual correction enough for a correctly rounded f32 quotient when the reciprocal is CORRECTLY ROUNDED?

    r = RN(1/d)  (once per ray and axis);   q0 = RN(x r);  e = RN(x - d q0) (one fma);  q1 = RN(q0 + e r) (one fma)   ==  RN(x/d) ?

The shipped FAST path uses the hardware divide's own chain: r = v_rcp refined once (NOT always the correctly rounded reciprocal) and
TWO corrections (five operations per quotient).  With r = RN(1/d) the exact value of q0 + e r is x/d (1 + theta), |theta| <= about 4 u^2
(u = 2^-24): q1 can differ from RN(x/d) only if x/d lies within that distance of a rounding boundary (a midpoint of two neighbouring
floats).  For significands X, D (integers in [2^23, 2^24)) and a midpoint Mo / 2^24 (Mo odd), the distance is |2^24 X - D Mo| / (2^24 D):
a non-zero integer N over 2^24 D.  So the only candidates are the pairs with |N| small -- and they can be ENUMERATED: for every D and every
small N, Mo = -N D^-1 mod 2^24 gives the one X.  This script checks all of them, in exact integer arithmetic, for both quotient binades
(X >= D and X < D), every D in [2^23, 2^24), |N| <= 8 (a D divisible by 16 admits no such N) -- plus random pairs, plus the same with a reciprocal one ulp off
(control: the enumeration must FIND failures there; random pairs do not).  Exponents do not matter (scaling by powers of two commutes with
every step) as long as nothing overflows or underflows -- which is what ray_is_nice / NTR_BVH_FASTDIV guarantee (trace_kernels.hip).

usage: div_one_correction_check.py [procs=8] [stride=1]      (stride > 1: every stride-th D, for a quick look)"""
import json
import multiprocessing as mp
import random
import sys

P = 24
LO, HI = 1 << (P - 1), 1 << P


def rn_ratio(num, den):
    """num / den (positive integers) rounded to nearest-even with P significant bits -> (mantissa in [2^(P-1), 2^P), exponent of two)"""
    sh = (den.bit_length() - num.bit_length()) + P + 1
    n2, d2 = (num << sh, den) if sh > 0 else (num, den << (-sh))
    q, r = divmod(n2, d2)
    extra = q.bit_length() - P
    low = q & ((1 << extra) - 1)
    q >>= extra
    half = 1 << (extra - 1)
    if low > half or (low == half and (r != 0 or (q & 1))):
        q += 1
    if q == (1 << P):
        q >>= 1
        extra += 1
    return q, extra - sh


def check(X, D, pert=0):
    """True if the one-correction chain gives RN(X / D); pert: the reciprocal moved by that many ulps (control)"""
    rm, re = rn_ratio(1, D)
    rm += pert
    q0m, q0e = rn_ratio(X * rm, 1)
    q0e += re
    s = -q0e
    eint = (X << s) - D * q0m            # x - d q0 = eint 2^-s, exactly
    if eint == 0:
        q1 = (q0m, q0e)
    else:
        sg = 1 if eint > 0 else -1
        em, ee = rn_ratio(abs(eint), 1)   # the fma rounds the residual (it is not always representable)
        ee -= s
        t2e = ee + re
        m = min(q0e, t2e)
        tot = (q0m << (q0e - m)) + sg * ((em * rm) << (t2e - m))
        q1m, q1e = rn_ratio(tot, 1)
        q1 = (q1m, q1e + m)
    return q1 == rn_ratio(X, D)


def candidates(D, nmax):
    """every X in [2^23, 2^24) whose quotient X / D lies within nmax / (2^24 D) (binade [1,2)) or nmax / (2^25 D) ([1/2,1)) of a midpoint"""
    k = (D & -D).bit_length() - 1
    Dp = D >> k
    for binade_bits, xlo, xhi in ((P, D, HI), (P + 1, LO, D)):
        mod = 1 << (binade_bits - k)
        inv = pow(Dp, -1, mod)
        for N in range(-nmax, nmax + 1):
            if N == 0 or (N & ((1 << k) - 1)):
                continue
            base = ((-(N >> k)) * inv) % mod
            for j in range(1 << k):
                Mo = base + j * mod
                while Mo < (1 << P):
                    Mo += (1 << binade_bits)
                if Mo >= (1 << (P + 1)) or not (Mo & 1):
                    continue
                X, rem = divmod(D * Mo + N, 1 << binade_bits)
                if rem == 0 and xlo <= X < xhi:
                    yield X


def work(args):
    d0, d1, stride, pert = args
    tested = bad = 0
    first = None
    for D in range(d0, d1):
        if D % 16 == 0 or (stride > 1 and ((D * 2654435761) >> 9) % stride):   # (a pseudo-random subset for stride > 1)
            continue
        for X in candidates(D, 8):
            tested += 1
            if not check(X, D, pert):
                bad += 1
                if first is None:
                    first = (X, D)
    return tested, bad, first


def main():
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    stride = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    chunks = [(d, min(d + (1 << 14), HI), stride, 0) for d in range(LO, HI, 1 << 14)]
    with mp.Pool(procs) as pool:
        res = pool.map(work, chunks)
        tested = sum(r[0] for r in res)
        bad = sum(r[1] for r in res)
        print(json.dumps(dict(what="near-midpoint quotients, correctly rounded reciprocal", stride=stride, pairs=tested, wrong=bad,
                              first=[r[2] for r in res if r[2]][:3])), flush=True)
        for pert in (1, -1):
            ctl = pool.map(work, [(d, min(d + (1 << 14), HI), max(stride, 1) * 64, pert) for d in range(LO, HI, 1 << 14)])
            print(json.dumps(dict(what="control: reciprocal off by %+d ulp" % pert, pairs=sum(r[0] for r in ctl), wrong=sum(r[1] for r in ctl))), flush=True)
    rnd = random.Random(7)
    badr = sum(not check(rnd.randrange(LO, HI), rnd.randrange(LO, HI)) for _ in range(1000000))
    print(json.dumps(dict(what="random pairs, correctly rounded reciprocal", pairs=1000000, wrong=badr)), flush=True)


if __name__ == "__main__":
    main()
