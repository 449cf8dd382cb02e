"""ML-KEM (FIPS 203, August 2024) written out from the standard's numbered
algorithms on Python integers and hashlib -- test-side only, slow, small cases.
An independent reading of the text next to oracle/kyber_oracle.c's ko_mlkem_*:
the two share no code (this one uses the standard's plain-domain NTT with
zeta = 17^BitRev7(i) and per-coefficient loops, the oracle the reference's
Montgomery arithmetic).  Neither is pinned by official vectors (none are
available offline): agreement shows a shared reading, not conformance."""
import hashlib

Q, N = 3329, 256
PARAMS = {2: (3, 2, 10, 4), 3: (2, 2, 10, 4), 4: (2, 2, 11, 5)}  # k: eta1, eta2, du, dv   (FIPS 203 table 2)


def bitrev7(i):
    return int("{:07b}".format(i)[::-1], 2)


ZETA = [pow(17, bitrev7(i), Q) for i in range(128)]          # algorithm 9/10
GAMMA = [pow(17, 2 * bitrev7(i) + 1, Q) for i in range(128)]  # algorithm 11


def H(b): return hashlib.sha3_256(b).digest()
def J(b): return hashlib.shake_256(b).digest(32)
def G(b):
    d = hashlib.sha3_512(b).digest()
    return d[:32], d[32:]
def PRF(eta, s, b): return hashlib.shake_256(s + bytes([b])).digest(64 * eta)


def bytes_to_bits(b):
    return [(x >> j) & 1 for x in b for j in range(8)]


def bits_to_bytes(bits):
    return bytes(sum(bits[8 * i + j] << j for j in range(8)) for i in range(len(bits) // 8))


def byte_encode(d, f):                                        # algorithm 5
    bits = []
    for a in f:
        bits += [(a >> j) & 1 for j in range(d)]
    return bits_to_bytes(bits)


def byte_decode(d, b):                                        # algorithm 6
    bits = bytes_to_bits(b)
    m = Q if d == 12 else 1 << d
    return [sum(bits[i * d + j] << j for j in range(d)) % m for i in range(N)]


def compress(d, x): return ((x << d) + Q // 2) // Q % (1 << d)      # 4.2.1, round(2^d/q * x) mod 2^d
def decompress(d, y): return (y * Q + (1 << (d - 1))) >> d            # round(q/2^d * y)


def sample_ntt(seed34):                                       # algorithm 7
    xof = hashlib.shake_128(seed34).digest(168 * 8)          # far more than ever needed
    a, i = [], 0
    while len(a) < N:
        c0, c1, c2 = xof[i], xof[i + 1], xof[i + 2]
        i += 3
        d1, d2 = c0 + 256 * (c1 % 16), c1 // 16 + 16 * c2
        if d1 < Q:
            a.append(d1)
        if d2 < Q and len(a) < N:
            a.append(d2)
    return a


def sample_cbd(eta, b):                                       # algorithm 8
    bits = bytes_to_bits(b)
    return [(sum(bits[2 * i * eta + j] for j in range(eta)) - sum(bits[2 * i * eta + eta + j] for j in range(eta))) % Q
            for i in range(N)]


def ntt(f):                                                   # algorithm 9
    f, i, ln = list(f), 1, 128
    while ln >= 2:
        for start in range(0, N, 2 * ln):
            z = ZETA[i]
            i += 1
            for j in range(start, start + ln):
                t = z * f[j + ln] % Q
                f[j + ln] = (f[j] - t) % Q
                f[j] = (f[j] + t) % Q
        ln //= 2
    return f


def ntt_inv(f):                                               # algorithm 10
    f, i, ln = list(f), 127, 2
    while ln <= 128:
        for start in range(0, N, 2 * ln):
            z = ZETA[i]
            i -= 1
            for j in range(start, start + ln):
                t = f[j]
                f[j] = (t + f[j + ln]) % Q
                f[j + ln] = z * (f[j + ln] - t) % Q
        ln *= 2
    return [x * 3303 % Q for x in f]


def mul_ntts(f, g):                                           # algorithms 11, 12
    h = [0] * N
    for i in range(128):
        a0, a1, b0, b1 = f[2 * i], f[2 * i + 1], g[2 * i], g[2 * i + 1]
        h[2 * i] = (a0 * b0 + a1 * b1 * GAMMA[i]) % Q
        h[2 * i + 1] = (a0 * b1 + a1 * b0) % Q
    return h


def add(f, g): return [(x + y) % Q for x, y in zip(f, g)]
def sub(f, g): return [(x - y) % Q for x, y in zip(f, g)]


def pke_keygen(k, d):                                         # algorithm 13
    eta1 = PARAMS[k][0]
    rho, sigma = G(d + bytes([k]))
    A = [[sample_ntt(rho + bytes([j, i])) for j in range(k)] for i in range(k)]
    s = [ntt(sample_cbd(eta1, PRF(eta1, sigma, n))) for n in range(k)]
    e = [ntt(sample_cbd(eta1, PRF(eta1, sigma, k + n))) for n in range(k)]
    t = []
    for i in range(k):
        acc = e[i]
        for j in range(k):
            acc = add(acc, mul_ntts(A[i][j], s[j]))
        t.append(acc)
    ek = b"".join(byte_encode(12, x) for x in t) + rho
    dk = b"".join(byte_encode(12, x) for x in s)
    return ek, dk


def pke_encrypt(k, ek, m, r):                                 # algorithm 14
    eta1, eta2, du, dv = PARAMS[k]
    t = [byte_decode(12, ek[384 * i:384 * i + 384]) for i in range(k)]
    rho = ek[384 * k:]
    A = [[sample_ntt(rho + bytes([j, i])) for j in range(k)] for i in range(k)]
    y = [ntt(sample_cbd(eta1, PRF(eta1, r, n))) for n in range(k)]
    e1 = [sample_cbd(eta2, PRF(eta2, r, k + n)) for n in range(k)]
    e2 = sample_cbd(eta2, PRF(eta2, r, 2 * k))
    u = []
    for i in range(k):
        acc = [0] * N
        for j in range(k):
            acc = add(acc, mul_ntts(A[j][i], y[j]))           # A^T
        u.append(add(ntt_inv(acc), e1[i]))
    mu = [decompress(1, b) for b in bytes_to_bits(m)]
    acc = [0] * N
    for j in range(k):
        acc = add(acc, mul_ntts(t[j], y[j]))
    v = add(add(ntt_inv(acc), e2), mu)
    c1 = b"".join(byte_encode(du, [compress(du, x) for x in ui]) for ui in u)
    return c1 + byte_encode(dv, [compress(dv, x) for x in v])


def pke_decrypt(k, dk, c):                                    # algorithm 15
    _, _, du, dv = PARAMS[k]
    u = [[decompress(du, x) for x in byte_decode(du, c[32 * du * i:32 * du * (i + 1)])] for i in range(k)]
    v = [decompress(dv, x) for x in byte_decode(dv, c[32 * du * k:])]
    s = [byte_decode(12, dk[384 * i:384 * i + 384]) for i in range(k)]
    acc = [0] * N
    for j in range(k):
        acc = add(acc, mul_ntts(s[j], ntt(u[j])))
    w = sub(v, ntt_inv(acc))
    return byte_encode(1, [compress(1, x) for x in w])


def keygen_internal(k, d, z):                                 # algorithm 16
    ek, dkp = pke_keygen(k, d)
    return ek, dkp + ek + H(ek) + z


def encaps_internal(k, ek, m):                                # algorithm 17
    K, r = G(m + H(ek))
    return K, pke_encrypt(k, ek, m, r)


def decaps_internal(k, dk, c):                                # algorithm 18
    dkp, ek = dk[:384 * k], dk[384 * k:768 * k + 32]
    h, z = dk[768 * k + 32:768 * k + 64], dk[768 * k + 64:]
    m = pke_decrypt(k, dkp, c)
    K, r = G(m + h)
    Kbar = J(z + c)
    return K if c == pke_encrypt(k, ek, m, r) else Kbar


def check_ek(k, ek):                                          # 7.2 modulus check
    return all(byte_encode(12, byte_decode(12, ek[384 * i:384 * i + 384])) == ek[384 * i:384 * i + 384] for i in range(k))
