"""Shared machinery for the golden-vector gates (test infrastructure).

* surf_coins(n): the byte strings the reference's test_vectors harness hands
  to the KEM, in order (ref/test_vectors.c:12-55): per iteration d, z, m0.
* tvec_stream(...): the exact stdout grammar of ref/test_vectors.c:57-105, so
  sha256 of 10 000 iterations can be compared with SHA256SUMS whatever
  produced pk/sk/ct/ss (oracle or GPU).
* NistDrbg: AES-256 CTR_DRBG of ref/rng.c:139-213 (AES from the oracle) and
  kat_record0(...): record 0 of PQCgenKAT_kem's .rsp (ref/PQCgenKAT_kem.c:50-105).
"""
import ctypes
import hashlib

import numpy as np

import oracle_lib

SETS = {  # name in SHA256SUMS / META -> (k, nineties)
    "tvecs512": (2, 0), "tvecs768": (3, 0), "tvecs1024": (4, 0),
    "tvecs512-90s": (2, 1), "tvecs768-90s": (3, 1), "tvecs1024-90s": (4, 1),
}
META_NAME = {(2, 0): "Kyber512", (3, 0): "Kyber768", (4, 0): "Kyber1024",
             (2, 1): "Kyber512-90s", (3, 1): "Kyber768-90s", (4, 1): "Kyber1024-90s"}


class _Surf(ctypes.Structure):
    _fields_ = [("inp", ctypes.c_uint32 * 12), ("out", ctypes.c_uint32 * 8), ("outleft", ctypes.c_int)]


def surf_coins(n):
    """-> (keypair_coins uint8[n,64], enc_coins uint8[n,32]) as test_vectors draws them."""
    lib = oracle_lib.load_oracle().lib
    s = _Surf()
    lib.ko_surf_init(ctypes.byref(s))
    buf = np.empty((n, 96), np.uint8)
    lib.ko_surf_bytes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    lib.ko_surf_bytes(ctypes.byref(s), buf.ctypes.data_as(ctypes.c_void_p), buf.size)
    return np.ascontiguousarray(buf[:, :64]), np.ascontiguousarray(buf[:, 64:])


def _hexlines(label, arr):
    """one b'label<hex>\\n' per row"""
    hx = np.frombuffer(b"0123456789abcdef", np.uint8)
    a = np.ascontiguousarray(arr, np.uint8)
    out = np.empty((a.shape[0], len(label) + 2 * a.shape[1] + 1), np.uint8)
    out[:, :len(label)] = np.frombuffer(label, np.uint8)
    out[:, len(label):-1:2] = hx[a >> 4]
    out[:, len(label) + 1:-1:2] = hx[a & 15]
    out[:, -1] = 10
    return out


def tvec_stream(kcoins, ecoins, pk, sk, ct, ssb, ssa):
    cols = [_hexlines(b"", kcoins[:, :32]), _hexlines(b"", kcoins[:, 32:]), _hexlines(b"Public Key: ", pk),
            _hexlines(b"Secret Key: ", sk), _hexlines(b"", ecoins), _hexlines(b"Ciphertext: ", ct),
            _hexlines(b"Shared Secret B: ", ssb), _hexlines(b"Shared Secret A: ", ssa)]
    return np.concatenate(cols, axis=1).tobytes()


def tvec_sha256(kcoins, ecoins, pk, sk, ct, ssb, ssa):
    return hashlib.sha256(tvec_stream(kcoins, ecoins, pk, sk, ct, ssb, ssa)).hexdigest()


def golden_sha256sums(path):
    out = {}
    with open(path) as fh:
        for line in fh:
            h, name = line.split()
            out[name] = h
    return out


class NistDrbg:
    """ref/rng.c:139-213 with the oracle's AES-256 as the block cipher."""

    def __init__(self, entropy48):
        self.lib = oracle_lib.load_oracle().lib
        self.key, self.v = bytes(32), bytes(16)
        self._update(bytes(entropy48))

    def _ecb(self, block):
        out = ctypes.create_string_buffer(16)
        self.lib.ko_aes256ctr(out, 16, self.key, block[:12], int.from_bytes(block[12:], "big"))
        return out.raw

    def _inc(self):
        self.v = ((int.from_bytes(self.v, "big") + 1) % (1 << 128)).to_bytes(16, "big")

    def _update(self, data):
        tmp = b""
        for _ in range(3):
            self._inc()
            tmp += self._ecb(self.v)
        if data is not None:
            tmp = bytes(a ^ b for a, b in zip(tmp, data))
        self.key, self.v = tmp[:32], tmp[32:]

    def randombytes(self, n):
        out = b""
        while len(out) < n:
            self._inc()
            out += self._ecb(self.v)
        self._update(None)
        return out[:n]


def kat_coins(count=1):
    """(seed48, d||z, m0) for the first `count` KAT records."""
    master = NistDrbg(bytes(range(48)))
    recs = []
    for _ in range(count):
        seed = master.randombytes(48)
        g = NistDrbg(seed)
        d, z, m0 = g.randombytes(32), g.randombytes(32), g.randombytes(32)
        recs.append((seed, d + z, m0))
    return recs


def kat_record(i, seed, pk, sk, ct, ss):
    def h(b):
        return bytes(b).hex().upper()
    return ("count = %d\nseed = %s\npk = %s\nsk = %s\nct = %s\nss = %s\n" % (i, h(seed), h(pk), h(sk), h(ct), h(ss))).encode()
