"""The FAST slab path's divide (one residual correction with the correctly rounded reciprocal) is exact because no quotient close to a
rounding boundary goes wrong -- scripts/studies/div_one_correction_check.py enumerates all of them (46.5 M significand pairs, 40 s on 8 cores;
profiles/r04_div_one_correction_check.txt).  Here: a pseudo-random 1/512 of the divisors, the control (a reciprocal one ulp off MUST fail on some
of the same pairs), and the integer rounding helper against exact rationals."""
import importlib.util
import os
import random
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("div_check", os.path.join(ROOT, "scripts", "studies", "div_one_correction_check.py"))
div_check = importlib.util.module_from_spec(spec)
spec.loader.exec_module(div_check)


def test_rounding_helper_equals_exact_rational_rounding():
    P = div_check.P

    def rn_frac(v):
        e = 0
        while v >= (1 << P):
            v /= 2
            e += 1
        while v < (1 << (P - 1)):
            v *= 2
            e -= 1
        fl = v.numerator // v.denominator
        rem = v - fl
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (fl & 1)):
            fl += 1
        if fl == (1 << P):
            fl >>= 1
            e += 1
        return fl, e
    rnd = random.Random(3)
    for _ in range(20000):
        a = rnd.randrange(1, 1 << rnd.randrange(1, 80))
        b = rnd.randrange(1, 1 << rnd.randrange(1, 60))
        if rnd.random() < 0.3:   # exact ties
            b = 1
            a = (rnd.randrange(1 << 23, 1 << 24) * 2 + 1) << rnd.randrange(0, 20)
        assert div_check.rn_ratio(a, b) == rn_frac(Fraction(a, b))


def test_one_correction_is_exact_on_a_subset_of_the_hardest_quotients_and_the_control_fails():
    tested = wrong = ctl_wrong = 0
    for d0 in range(div_check.LO, div_check.HI, 1 << 18):
        t, b, _ = div_check.work((d0, d0 + (1 << 18), 512, 0))
        tested += t
        wrong += b
        ctl_wrong += div_check.work((d0, d0 + (1 << 18), 512, 1))[1]
    assert tested > 50000
    assert wrong == 0
    assert ctl_wrong > 0.05 * tested      # the enumeration really is where a worse reciprocal breaks
    rnd = random.Random(11)
    assert all(div_check.check(rnd.randrange(div_check.LO, div_check.HI), rnd.randrange(div_check.LO, div_check.HI)) for _ in range(20000))
