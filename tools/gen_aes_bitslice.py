#!/usr/bin/env python3
"""Derive a bitsliced AES S-box circuit from first principles and emit it as a C++ header
(tools/microbench/aes_bs_sbox.h) for the bitsliced variant of tools/microbench/aes_rate.hip.

Nothing is copied from a published gate list: the S-box is inversion in GF(2^8) followed by the
affine map (FIPS 197 5.1.1); the inversion is computed in the tower GF(((2^2)^2)^2), where it costs
a handful of GF(2^4) products, and two linear maps (the field isomorphism in, the isomorphism out
merged with the affine map) connect it to the AES representation.  The circuit is built once over
symbolic bits with hash-consing (so common subexpressions are shared), checked against the S-box on
all 256 inputs, and written out as straight-line xor / and / not code on word-sized bit planes.

    python3 tools/gen_aes_bitslice.py          # rewrites tools/microbench/aes_bs_sbox.h
"""
import itertools
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- circuit builder
class Net:
    """nodes: ('in', i) | ('xor', a, b) | ('and', a, b) | ('not', a); constants are the ints 0 / 1"""

    def __init__(self):
        self.nodes, self.index = [], {}

    def _mk(self, key):
        if key not in self.index:
            self.index[key] = len(self.nodes)
            self.nodes.append(key)
        return Bit(self, self.index[key])

    def inp(self, i):
        return self._mk(("in", i))

    def xor(self, a, b):
        if isinstance(a, int) and isinstance(b, int):
            return a ^ b
        if isinstance(a, int):
            a, b = b, a
        if isinstance(b, int):
            return a if b == 0 else self._mk(("not", a.id))
        if a.id == b.id:
            return 0
        return self._mk(("xor",) + tuple(sorted((a.id, b.id))))

    def land(self, a, b):
        if isinstance(a, int) and isinstance(b, int):
            return a & b
        if isinstance(a, int):
            a, b = b, a
        if isinstance(b, int):
            return a if b else 0
        if a.id == b.id:
            return a
        return self._mk(("and",) + tuple(sorted((a.id, b.id))))


class Bit:
    def __init__(self, net, id_):
        self.net, self.id = net, id_

    def __xor__(self, o):
        return self.net.xor(self, o)

    __rxor__ = __xor__

    def __and__(self, o):
        return self.net.land(self, o)

    __rand__ = __and__


# ---------------------------------------------------------------- tower field, generic over ints and Bits
# GF(4) = GF(2)[u]/(u^2+u+1), elements (h, l) = h u + l
def mul4(a, b):
    t = (a[0] ^ a[1]) & (b[0] ^ b[1])
    p, q = a[1] & b[1], a[0] & b[0]
    return (t ^ p, p ^ q)


def sq4(a):
    return (a[0], a[0] ^ a[1])


def scale4(a):  # times u
    return (a[0] ^ a[1], a[0])


def add4(a, b):
    return (a[0] ^ b[0], a[1] ^ b[1])


# GF(16) = GF(4)[v]/(v^2+v+u), elements (H, L)
def mul16(a, b):
    p = mul4(add4(a[0], a[1]), add4(b[0], b[1]))
    q, r = mul4(a[0], b[0]), mul4(a[1], b[1])
    return (add4(p, r), add4(r, scale4(q)))


def sq16(a):
    h = sq4(a[0])
    return (h, add4(scale4(h), sq4(a[1])))


def add16(a, b):
    return (add4(a[0], b[0]), add4(a[1], b[1]))


def inv16(a):
    d = add4(add4(scale4(sq4(a[0])), mul4(a[0], a[1])), sq4(a[1]))
    di = sq4(d)  # x^-1 = x^2 in GF(4)
    return (mul4(a[0], di), mul4(add4(a[0], a[1]), di))


def const16(v):
    return (((v >> 3) & 1, (v >> 2) & 1), ((v >> 1) & 1, v & 1))


def val16(a):
    return a[0][0] << 3 | a[0][1] << 2 | a[1][0] << 1 | a[1][1]


def find_m():
    """M in GF(16) with w^2 + w + M irreducible: no root in GF(16)"""
    for m in range(1, 16):
        if all(val16(add16(add16(sq16(const16(x)), const16(x)), const16(m))) != 0 for x in range(16)):
            return m
    raise AssertionError


M16 = find_m()


# GF(256) = GF(16)[w]/(w^2+w+M), elements (H, L)
def mul256(a, b):
    p = mul16(add16(a[0], a[1]), add16(b[0], b[1]))
    q, r = mul16(a[0], b[0]), mul16(a[1], b[1])
    return (add16(p, r), add16(r, mul16(q, const16(M16))))


def inv256(a):
    d = add16(add16(mul16(sq16(a[0]), const16(M16)), mul16(a[0], a[1])), sq16(a[1]))
    di = inv16(d)
    return (mul16(a[0], di), mul16(add16(a[0], a[1]), di))


def const256(v):
    return (const16(v >> 4), const16(v & 15))


def val256(a):
    return val16(a[0]) << 4 | val16(a[1])


def bits256(a):
    """tower element -> list of its 8 bits, bit 7 first"""
    return [a[0][0][0], a[0][0][1], a[0][1][0], a[0][1][1], a[1][0][0], a[1][0][1], a[1][1][0], a[1][1][1]]


def from_bits(b):
    return (((b[0], b[1]), (b[2], b[3])), ((b[4], b[5]), (b[6], b[7])))


# ---------------------------------------------------------------- AES side
def aes_mul(a, b):
    r = 0
    while b:
        if b & 1:
            r ^= a
        a = ((a << 1) ^ 0x11B) if a & 0x80 else a << 1
        b >>= 1
    return r


def aes_sbox():
    out = []
    for a in range(256):
        x = next((b for b in range(1, 256) if aes_mul(a, b) == 1), 0)
        y = x
        for s in (1, 2, 3, 4):
            y ^= ((x << s) | (x >> (8 - s))) & 0xFF
        out.append(y ^ 0x63)
    return out


def isomorphism():
    """8x8 GF(2) matrices (lists of 8 column bytes): AES polynomial basis -> tower, and back"""
    one = const256(1)
    for beta in range(2, 256):
        b = const256(beta)
        pw = [one]
        for _ in range(8):
            pw.append(mul256(pw[-1], b))
        acc = 0
        for e in (8, 4, 3, 1, 0):  # x^8 + x^4 + x^3 + x + 1
            acc ^= val256(pw[e])
        if acc == 0:
            cols = [val256(pw[i]) for i in range(8)]  # image of x^i
            break
    else:
        raise AssertionError("no root of the AES polynomial in the tower field")

    def apply(cols_, v):
        r = 0
        for i in range(8):
            if (v >> i) & 1:
                r ^= cols_[i]
        return r
    inv_cols = [next(v for v in range(256) if apply(cols, v) == 1 << i) for i in range(8)]
    # homomorphism check on products
    for a, b in itertools.product((3, 0x53, 0xCA, 0xFF), (2, 0x11, 0x80)):
        assert apply(inv_cols, val256(mul256(const256(apply(cols, a)), const256(apply(cols, b))))) == aes_mul(a, b)
    return cols, inv_cols


def linear(net_bits, cols, nout=8):
    """out bit i = xor of in bits j with bit i of cols[j] set; in/out lists are bit 0 first"""
    out = []
    for i in range(nout):
        acc = 0
        for j in range(8):
            if (cols[j] >> i) & 1:
                acc = acc ^ net_bits[j]
        out.append(acc)
    return out


def build():
    net = Net()
    x = [net.inp(i) for i in range(8)]  # bit 0 first
    to_tower, from_tower = isomorphism()
    t = linear(x, to_tower)
    y = inv256(from_bits(t[::-1]))
    yb = bits256(y)[::-1]
    # affine map of FIPS 197 on the AES-basis bits, merged with the map back: columns of A * from_tower
    def affine(v):
        r = v
        for s in (1, 2, 3, 4):
            r ^= ((v << s) | (v >> (8 - s))) & 0xFF
        return r
    cols = [affine(c) for c in from_tower]
    s = linear(yb, cols)
    s = [s[i] ^ ((0x63 >> i) & 1) for i in range(8)]
    return net, s


def evaluate(net, outs):
    """all 256 inputs at once: a node's value is a 256-bit integer, bit t = its value on input t"""
    full = (1 << 256) - 1
    val = []
    for node in net.nodes:
        if node[0] == "in":
            v = sum(1 << t for t in range(256) if (t >> node[1]) & 1)
        elif node[0] == "xor":
            v = val[node[1]] ^ val[node[2]]
        elif node[0] == "and":
            v = val[node[1]] & val[node[2]]
        else:
            v = val[node[1]] ^ full
        val.append(v)
    return [val[o.id] for o in outs]


def emit(net, outs, path):
    used, stack = set(), [o.id for o in outs]
    while stack:
        n = stack.pop()
        if n in used:
            continue
        used.add(n)
        stack.extend(net.nodes[n][1:] if net.nodes[n][0] != "in" else [])
    counts = {"xor": 0, "and": 0, "not": 0}
    name = {}
    lines = []
    for n in sorted(used):
        node = net.nodes[n]
        if node[0] == "in":
            name[n] = "x[%d]" % node[1]
            continue
        counts[node[0]] += 1
        name[n] = "t%d" % n
        if node[0] == "not":
            lines.append("  const W t%d = ~%s;" % (n, name[node[1]]))
        else:
            lines.append("  const W t%d = %s %s %s;" % (n, name[node[1]], "^" if node[0] == "xor" else "&", name[node[2]]))
    with open(path, "w") as fh:
        fh.write("// GENERATED by tools/gen_aes_bitslice.py -- do not edit.\n")
        fh.write("// AES S-box on bit planes: x[i] holds bit i of the S-box input of every slice, replaced by the output.\n")
        fh.write("// Inversion in GF(((2^2)^2)^2) (w^2+w+%d over v^2+v+u over u^2+u+1) between two linear maps;\n" % M16)
        fh.write("// %d xor, %d and, %d not = %d gates, checked against the S-box on all 256 inputs by the generator.\n"
                 % (counts["xor"], counts["and"], counts["not"], sum(counts.values())))
        fh.write("#pragma once\n#define KY_BS_SBOX_GATES %d\ntemplate <class W>\n__device__ __forceinline__ void bs_sbox(W (&x)[8]) {\n" % sum(counts.values()))
        fh.write("\n".join(lines) + "\n")
        for i, o in enumerate(outs):
            fh.write("  x[%d] = %s;\n" % (i, name[o.id]))
        fh.write("}\n")
    return counts


def main():
    net, outs = build()
    got = evaluate(net, outs)
    sbox = aes_sbox()
    assert sbox[0] == 0x63 and sbox[1] == 0x7C and sbox[0x53] == 0xED
    for i in range(8):
        want = sum(1 << t for t in range(256) if (sbox[t] >> i) & 1)
        assert got[i] == want, "output bit %d differs from the S-box" % i
    counts = emit(net, outs, os.path.join(ROOT, "tools", "microbench", "aes_bs_sbox.h"))
    print("wrote tools/microbench/aes_bs_sbox.h:", counts, "M =", M16)


if __name__ == "__main__":
    main()
