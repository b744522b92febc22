"""ORACLE (test infrastructure only): BN254 optimal-ate pairing in plain Python integers.

Used only by the oracle verifier (oracle/halo2_ref.py) to check KZG/SHPLONK openings: the reference's
verifier is halo2_proofs `verify_proof` + halo2curves' pairing (third-party, not on disk; reference
README.md:48-52).  Textbook construction: Fq12 = Fq[w]/(w^12 - 18 w^6 + 82), G2 on the sextic twist
y^2 = x^3 + 3/(9+i), Miller loop over 6t+2 with the two Frobenius corrections, final exponentiation by
(q^12-1)/r.  Slow (seconds) and simple on purpose.
"""
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
ATE_LOOP_COUNT = 29793968203157093288  # 6t + 2, t = 4965661367192848881
LOG_ATE = 63
FQ12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]  # w^12 = 18 w^6 - 82


class FQP:
    """element of Fq[w]/(w^deg - sum mod_coeffs[i] w^i ... ) given as coefficient list"""
    degree = 0
    mod = None

    def __init__(self, coeffs):
        self.c = [x % Q for x in coeffs]

    def __add__(self, o):
        return self.__class__([a + b for a, b in zip(self.c, o.c)])

    def __sub__(self, o):
        return self.__class__([a - b for a, b in zip(self.c, o.c)])

    def __neg__(self):
        return self.__class__([-a for a in self.c])

    def __eq__(self, o):
        return self.c == o.c

    def scale(self, k):
        return self.__class__([a * k for a in self.c])

    def __mul__(self, o):
        if isinstance(o, int):
            return self.scale(o)
        d = self.degree
        b = [0] * (2 * d - 1)
        for i, x in enumerate(self.c):
            if x:
                for j, y in enumerate(o.c):
                    b[i + j] += x * y
        m = self.mod
        for exp in range(2 * d - 2, d - 1, -1):
            top = b[exp] % Q
            b[exp] = 0
            if top:
                for i in range(d):
                    if m[i]:
                        b[exp - d + i] -= top * m[i]
        return self.__class__(b[:d])

    def inv(self):
        # extended Euclid on polynomials
        d = self.degree
        lm, hm = [1] + [0] * d, [0] * (d + 1)
        low, high = self.c + [0], list(self.mod) + [1]

        def deg(p):
            k = len(p) - 1
            while k and p[k] % Q == 0:
                k -= 1
            return k

        def poly_rounded_div(a, b):
            dega, degb = deg(a), deg(b)
            temp = list(a)
            o = [0] * len(a)
            for i in range(dega - degb, -1, -1):
                o[i] = (o[i] + temp[degb + i] * pow(b[degb], -1, Q)) % Q
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - o[c]) % Q if False else temp[c + i]
                # subtract o[i] * b * w^i
                for c in range(degb + 1):
                    temp[c + i] = (temp[c + i] - o[i] * b[c]) % Q
            return o[: deg(o) + 1]

        while deg(low):
            r = poly_rounded_div(high, low)
            r += [0] * (d + 1 - len(r))
            nm, new = list(hm), list(high)
            for i in range(d + 1):
                for j in range(d + 1 - i):
                    nm[i + j] -= lm[i] * r[j]
                    new[i + j] -= low[i] * r[j]
            nm = [x % Q for x in nm]
            new = [x % Q for x in new]
            lm, low, hm, high = nm, new, lm, low
        inv0 = pow(low[0], -1, Q)
        return self.__class__([x * inv0 for x in lm[:d]])

    def __pow__(self, e):
        out = self.one()
        base = self
        while e:
            if e & 1:
                out = out * base
            base = base * base
            e >>= 1
        return out

    @classmethod
    def one(cls):
        return cls([1] + [0] * (cls.degree - 1))

    @classmethod
    def zero(cls):
        return cls([0] * cls.degree)

    def is_zero(self):
        return all(x == 0 for x in self.c)


class FQ2(FQP):
    degree = 2
    mod = [1, 0]  # i^2 = -1


class FQ12(FQP):
    degree = 12
    mod = FQ12_MOD


# twist curve b2 = 3 / (9 + i)
B2 = FQ2([3, 0]) * FQ2([9, 1]).inv()
B12 = FQ12([3] + [0] * 11)

G2_GEN = (FQ2([10857046999023057135944570762232829481370756359578518086990519993285655852781,
               11559732032986387107991004021392285783925812861821192530917403151452391805634]),
          FQ2([8495653923123431417604973247489272438418190587263600148770280649306958101930,
               4082367875863433681332203403145435568316851327593401208105741076214120093531]))


def ec_double(P):
    if P is None:
        return None
    x, y = P
    if y.is_zero():
        return None
    lam = (x * x * 3) * (y * 2).inv()
    nx = lam * lam - x * 2
    ny = lam * (x - nx) - y
    return (nx, ny)


def ec_add(P1, P2):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    x1, y1 = P1
    x2, y2 = P2
    if x1 == x2:
        if y1 == y2:
            return ec_double(P1)
        return None
    lam = (y2 - y1) * (x2 - x1).inv()
    nx = lam * lam - x1 - x2
    ny = lam * (x1 - nx) - y1
    return (nx, ny)


def ec_mul(P, k):
    acc = None
    while k:
        if k & 1:
            acc = ec_add(acc, P)
        P = ec_double(P)
        k >>= 1
    return acc


def ec_neg(P):
    return None if P is None else (P[0], -P[1])


def g2_on_curve(P):
    if P is None:
        return True
    x, y = P
    return y * y - x * x * x == B2


W = FQ12([0, 1] + [0] * 10)


def twist(P):
    """G2 point on the twist over Fq2 -> point on y^2 = x^3 + 3 over Fq12"""
    if P is None:
        return None
    x, y = P
    xc = [x.c[0] - x.c[1] * 9, x.c[1]]
    yc = [y.c[0] - y.c[1] * 9, y.c[1]]
    nx = FQ12([xc[0]] + [0] * 5 + [xc[1]] + [0] * 5)
    ny = FQ12([yc[0]] + [0] * 5 + [yc[1]] + [0] * 5)
    return (nx * (W ** 2), ny * (W ** 3))


def cast_g1(P):
    x, y = P
    return (FQ12([x] + [0] * 11), FQ12([y] + [0] * 11))


def linefunc(P1, P2, T):
    x1, y1 = P1
    x2, y2 = P2
    xt, yt = T
    if not (x1 == x2):
        m = (y2 - y1) * (x2 - x1).inv()
        return m * (xt - x1) - (yt - y1)
    elif y1 == y2:
        m = (x1 * x1 * 3) * (y1 * 2).inv()
        return m * (xt - x1) - (yt - y1)
    else:
        return xt - x1


def frob_point(P):
    return (P[0] ** Q, P[1] ** Q)


def miller_loop(Qt, Pt):
    if Qt is None or Pt is None:
        return FQ12.one()
    Rp = Qt
    f = FQ12.one()
    for i in range(LOG_ATE, -1, -1):
        f = f * f * linefunc(Rp, Rp, Pt)
        Rp = ec_double(Rp)
        if ATE_LOOP_COUNT & (1 << i):
            f = f * linefunc(Rp, Qt, Pt)
            Rp = ec_add(Rp, Qt)
    Q1 = frob_point(Qt)
    nQ2 = ec_neg(frob_point(Q1))
    f = f * linefunc(Rp, Q1, Pt)
    Rp = ec_add(Rp, Q1)
    f = f * linefunc(Rp, nQ2, Pt)
    return f


def final_exp(f):
    return f ** ((Q ** 12 - 1) // R)


def pairing(Q2, P1):
    """e(P1 in G1 (ints x,y or None), Q2 in G2 (FQ2 pair or None)) -> Fq12"""
    if P1 is None or Q2 is None:
        return FQ12.one()
    return final_exp(miller_loop(twist(Q2), cast_g1(P1)))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1, one shared final exponentiation"""
    f = FQ12.one()
    for P1, Q2 in pairs:
        if P1 is None or Q2 is None:
            continue
        f = f * miller_loop(twist(Q2), cast_g1(P1))
    return final_exp(f) == FQ12.one()
