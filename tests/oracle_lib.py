"""Test-side loader for the CHECKERS: oracle/libkyber_oracle.so (the CPU
restatement) and oracle/_ref/*.so (the reference's own sources compiled in
place by oracle/Makefile).  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module; the product never does."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


class KoParams(ctypes.Structure):
    _fields_ = [("k", ctypes.c_int), ("eta1", ctypes.c_int), ("du", ctypes.c_int), ("dv", ctypes.c_int),
                ("nineties", ctypes.c_int)]


def build_oracle():
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "-j8", "all"], check=True)


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class _Kex:
    """ref/kex.c's six functions over dense per-instance arrays (instance i at
    row i), driven one instance at a time through `call(name, *row_pointers)`."""

    def __init__(self, call, pkb, skb, ctb):
        self.call, self.pkb, self.skb, self.ctb = call, pkb, skb, ctb

    @staticmethod
    def _rows(a, width):
        return np.ascontiguousarray(a, np.uint8).reshape(-1, width)

    def _run(self, name, outs, ins):
        n = ins[0].shape[0]
        outs = [np.zeros((n, w), np.uint8) for w in outs]
        for i in range(n):
            self.call(name, *[_p(o[i]) for o in outs], *[_p(a[i]) for a in ins])
        return outs

    def uake_initA(self, pkb, coins):
        return self._run("uake_initA", [self.pkb + self.ctb, 32, self.skb], [self._rows(pkb, self.pkb), self._rows(coins, 96)])

    def ake_initA(self, pkb, coins):
        return self._run("ake_initA", [self.pkb + self.ctb, 32, self.skb], [self._rows(pkb, self.pkb), self._rows(coins, 96)])

    def uake_sharedB(self, recv, skb, coins):
        return self._run("uake_sharedB", [self.ctb, 32],
                         [self._rows(recv, self.pkb + self.ctb), self._rows(skb, self.skb), self._rows(coins, 32)])

    def uake_sharedA(self, recv, tk, sk):
        return self._run("uake_sharedA", [32], [self._rows(recv, self.ctb), self._rows(tk, 32), self._rows(sk, self.skb)])[0]

    def ake_sharedB(self, recv, skb, pka, coins):
        return self._run("ake_sharedB", [2 * self.ctb, 32],
                         [self._rows(recv, self.pkb + self.ctb), self._rows(skb, self.skb), self._rows(pka, self.pkb),
                          self._rows(coins, 64)])

    def ake_sharedA(self, recv, tk, sk, ska):
        return self._run("ake_sharedA", [32], [self._rows(recv, 2 * self.ctb), self._rows(tk, 32), self._rows(sk, self.skb),
                                               self._rows(ska, self.skb)])[0]


class Oracle:
    def kex(self, k, nineties=0):
        """-> _Kex over the oracle's ko_kex_* (ref/kex.c restated)."""
        p = self.params(k, nineties)
        pkb, skb, ctb = self.sizes(k)

        def call(name, *ptrs):
            fn = getattr(self.lib, "ko_kex_" + name)
            fn.restype = None
            fn(ctypes.byref(p), *ptrs)
        return _Kex(call, pkb, skb, ctb)

    def __init__(self, lib):
        self.lib = lib
        vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
        P = ctypes.POINTER(KoParams)
        lib.ko_params_init.argtypes = [P, ci, ci]
        for name in ("ko_pk_bytes", "ko_sk_bytes", "ko_ct_bytes"):
            getattr(lib, name).argtypes = [P]
            getattr(lib, name).restype = sz
        lib.ko_batch_keypair.argtypes = [P, sz, vp, vp, vp]
        lib.ko_batch_enc.argtypes = [P, sz, vp, vp, vp, vp]
        lib.ko_batch_dec.argtypes = [P, sz, vp, vp, vp]
        lib.ko_gen_matrix.argtypes = [P, vp, vp, ci, vp]
        lib.ko_getnoise.argtypes = [P, ci, vp, vp, ctypes.c_uint8]
        for name in ("ko_ntt", "ko_poly_ntt", "ko_invntt_tomont", "ko_poly_tomont"):
            getattr(lib, name).argtypes = [vp]
        lib.ko_poly_basemul.argtypes = [vp, vp, vp]
        lib.ko_polyvec_basemul_acc.argtypes = [ci, vp, vp, vp]
        lib.ko_cbd.argtypes = [ci, vp, vp]
        lib.ko_poly_tobytes.argtypes = [vp, vp]
        lib.ko_poly_frombytes.argtypes = [vp, vp]
        lib.ko_poly_compress.argtypes = [ci, vp, vp]
        lib.ko_poly_decompress.argtypes = [ci, vp, vp]
        lib.ko_poly_frommsg.argtypes = [vp, vp]
        lib.ko_poly_tomsg.argtypes = [vp, vp]
        lib.ko_zetas.restype = ctypes.POINTER(ctypes.c_int16)
        lib.ko_montgomery_reduce.argtypes = [ctypes.c_int32]
        lib.ko_montgomery_reduce.restype = ctypes.c_int16
        lib.ko_barrett_reduce.argtypes = [ctypes.c_int16]
        lib.ko_barrett_reduce.restype = ctypes.c_int16
        for name in ("ko_shake128", "ko_shake256"):
            getattr(lib, name).argtypes = [vp, sz, vp, sz]
        for name in ("ko_sha3_256", "ko_sha3_512", "ko_sha256", "ko_sha512"):
            getattr(lib, name).argtypes = [vp, vp, sz]
        lib.ko_aes256ctr.argtypes = [vp, sz, vp, vp, ctypes.c_uint32]
        lib.ko_keccak_f1600.argtypes = [vp]

    def params(self, k, nineties=0):
        p = KoParams()
        if self.lib.ko_params_init(ctypes.byref(p), k, int(nineties)):
            raise ValueError("bad k")
        return p

    def sizes(self, k):
        p = self.params(k)
        return (self.lib.ko_pk_bytes(ctypes.byref(p)), self.lib.ko_sk_bytes(ctypes.byref(p)),
                self.lib.ko_ct_bytes(ctypes.byref(p)))

    def keypair(self, k, coins, nineties=0):
        p = self.params(k, nineties)
        coins = np.ascontiguousarray(coins, np.uint8).reshape(-1, 64)
        n = coins.shape[0]
        pkb, skb, _ = self.sizes(k)
        pk, sk = np.empty((n, pkb), np.uint8), np.empty((n, skb), np.uint8)
        self.lib.ko_batch_keypair(ctypes.byref(p), n, _p(pk), _p(sk), _p(coins))
        return pk, sk

    def enc(self, k, pk, coins, nineties=0):
        p = self.params(k, nineties)
        pkb, _, ctb = self.sizes(k)
        pk = np.ascontiguousarray(pk, np.uint8).reshape(-1, pkb)
        coins = np.ascontiguousarray(coins, np.uint8).reshape(-1, 32)
        n = pk.shape[0]
        ct, ss = np.empty((n, ctb), np.uint8), np.empty((n, 32), np.uint8)
        self.lib.ko_batch_enc(ctypes.byref(p), n, _p(ct), _p(ss), _p(pk), _p(coins))
        return ct, ss

    def dec(self, k, ct, sk, nineties=0):
        p = self.params(k, nineties)
        _, skb, ctb = self.sizes(k)
        ct = np.ascontiguousarray(ct, np.uint8).reshape(-1, ctb)
        sk = np.ascontiguousarray(sk, np.uint8).reshape(-1, skb)
        n = ct.shape[0]
        ss = np.empty((n, 32), np.uint8)
        self.lib.ko_batch_dec(ctypes.byref(p), n, _p(ss), _p(ct), _p(sk))
        return ss

    # ---- ML-KEM (FIPS 203) internal algorithms; parity unpinned, see oracle/kyber_oracle.c ----
    def mlkem_keypair(self, k, coins):
        p = self.params(k)
        coins = np.ascontiguousarray(coins, np.uint8).reshape(-1, 64)
        n = coins.shape[0]
        pkb, skb, _ = self.sizes(k)
        ek, dk = np.empty((n, pkb), np.uint8), np.empty((n, skb), np.uint8)
        self.lib.ko_mlkem_batch_keypair(ctypes.byref(p), ctypes.c_size_t(n), _p(ek), _p(dk), _p(coins))
        return ek, dk

    def mlkem_enc(self, k, ek, coins):
        p = self.params(k)
        pkb, _, ctb = self.sizes(k)
        ek = np.ascontiguousarray(ek, np.uint8).reshape(-1, pkb)
        coins = np.ascontiguousarray(coins, np.uint8).reshape(-1, 32)
        n = ek.shape[0]
        ct, ss = np.empty((n, ctb), np.uint8), np.empty((n, 32), np.uint8)
        self.lib.ko_mlkem_batch_enc(ctypes.byref(p), ctypes.c_size_t(n), _p(ct), _p(ss), _p(ek), _p(coins))
        return ct, ss

    def mlkem_dec(self, k, ct, dk):
        p = self.params(k)
        _, skb, ctb = self.sizes(k)
        ct = np.ascontiguousarray(ct, np.uint8).reshape(-1, ctb)
        dk = np.ascontiguousarray(dk, np.uint8).reshape(-1, skb)
        n = ct.shape[0]
        ss = np.empty((n, 32), np.uint8)
        self.lib.ko_mlkem_batch_dec(ctypes.byref(p), ctypes.c_size_t(n), _p(ss), _p(ct), _p(dk))
        return ss

    def gen_matrix(self, k, seed, transposed, nineties=0):
        p = self.params(k, nineties)
        seed = np.ascontiguousarray(seed, np.uint8)
        a = np.empty((k * k, 256), np.int16)
        nb = np.zeros(k * k, np.int32)
        self.lib.ko_gen_matrix(ctypes.byref(p), _p(a), _p(seed), int(transposed), _p(nb))
        return a, nb

    def poly_op(self, name, r):
        r = np.ascontiguousarray(r, np.int16).copy()
        fn = getattr(self.lib, name)
        for row in r.reshape(-1, 256):
            fn(_p(row))
        return r

    def getnoise(self, k, nineties, eta, seed, nonce):
        p = self.params(k, nineties)
        seed = np.ascontiguousarray(seed, np.uint8)
        r = np.empty(256, np.int16)
        self.lib.ko_getnoise(ctypes.byref(p), int(eta), _p(r), _p(seed), int(nonce))
        return r

    def cbd(self, eta, buf):
        buf = np.ascontiguousarray(buf, np.uint8).reshape(-1, 64 * eta)
        r = np.empty((buf.shape[0], 256), np.int16)
        for i in range(buf.shape[0]):
            self.lib.ko_cbd(int(eta), _p(r[i]), _p(buf[i]))
        return r

    def codec(self, name, src, out_shape, out_dtype, d=None):
        """ko_poly_{tobytes,frombytes,compress,decompress,frommsg,tomsg} row by row."""
        src = np.ascontiguousarray(src)
        out = np.empty((src.shape[0],) + out_shape, out_dtype)
        fn = getattr(self.lib, name)
        for i in range(src.shape[0]):
            if d is None:
                fn(_p(out[i]), _p(src[i]))
            else:
                fn(int(d), _p(out[i]), _p(src[i]))
        return out

    def polyvec_basemul_acc(self, k, a, b):
        a = np.ascontiguousarray(a, np.int16).reshape(-1, k, 256)
        b = np.ascontiguousarray(b, np.int16).reshape(-1, k, 256)
        r = np.empty((a.shape[0], 256), np.int16)
        for i in range(a.shape[0]):
            self.lib.ko_polyvec_basemul_acc(int(k), _p(r[i]), _p(a[i]), _p(b[i]))
        return r

    def poly_basemul(self, a, b):
        a = np.ascontiguousarray(a, np.int16).reshape(-1, 256)
        b = np.ascontiguousarray(b, np.int16).reshape(-1, 256)
        r = np.empty_like(a)
        for i in range(a.shape[0]):
            self.lib.ko_poly_basemul(_p(r[i]), _p(a[i]), _p(b[i]))
        return r


_ORACLE = None


def load_oracle():
    global _ORACLE
    if _ORACLE is None:
        path = os.path.join(ORACLE_DIR, "libkyber_oracle.so")
        if not os.path.exists(path):
            build_oracle()
        _ORACLE = Oracle(ctypes.CDLL(path))
    return _ORACLE


REF_SETS = {(2, 0): "512", (3, 0): "768", (4, 0): "1024", (2, 1): "512_90s", (3, 1): "768_90s", (4, 1): "1024_90s"}


class RefLib:
    """One parameter set of the reference (impl = 'ref' or 'avx2') via oracle/ref_shim.c."""

    def __init__(self, impl, k, nineties=0):
        path = os.path.join(ORACLE_DIR, "_ref", "libkyber_%s_%s.so" % (impl, REF_SETS[(k, int(nineties))]))
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = ctypes.CDLL(path)
        vp, sz = ctypes.c_void_p, ctypes.c_size_t
        for name in ("refshim_pk_bytes", "refshim_sk_bytes", "refshim_ct_bytes"):
            getattr(self.lib, name).restype = sz
        self.lib.refshim_batch_keypair.argtypes = [sz, vp, vp, vp]
        self.lib.refshim_batch_enc.argtypes = [sz, vp, vp, vp, vp]
        self.lib.refshim_batch_dec.argtypes = [sz, vp, vp, vp]
        self.pkb, self.skb, self.ctb = (self.lib.refshim_pk_bytes(), self.lib.refshim_sk_bytes(),
                                        self.lib.refshim_ct_bytes())

    def kex(self):
        """-> _Kex over the reference's own kex.c (through the shim's coin cursor)."""
        def call(name, *ptrs):
            fn = getattr(self.lib, "refshim_kex_" + name)
            fn.restype = None
            fn(*ptrs)
        return _Kex(call, self.pkb, self.skb, self.ctb)

    def keypair(self, coins):
        coins = np.ascontiguousarray(coins, np.uint8).reshape(-1, 64)
        n = coins.shape[0]
        pk, sk = np.empty((n, self.pkb), np.uint8), np.empty((n, self.skb), np.uint8)
        self.lib.refshim_batch_keypair(n, _p(pk), _p(sk), _p(coins))
        return pk, sk

    def enc(self, pk, coins):
        pk = np.ascontiguousarray(pk, np.uint8).reshape(-1, self.pkb)
        coins = np.ascontiguousarray(coins, np.uint8).reshape(-1, 32)
        n = pk.shape[0]
        ct, ss = np.empty((n, self.ctb), np.uint8), np.empty((n, 32), np.uint8)
        self.lib.refshim_batch_enc(n, _p(ct), _p(ss), _p(pk), _p(coins))
        return ct, ss

    def dec(self, ct, sk):
        ct = np.ascontiguousarray(ct, np.uint8).reshape(-1, self.ctb)
        sk = np.ascontiguousarray(sk, np.uint8).reshape(-1, self.skb)
        n = ct.shape[0]
        ss = np.empty((n, 32), np.uint8)
        self.lib.refshim_batch_dec(n, _p(ss), _p(ct), _p(sk))
        return ss


def has_avx2():
    try:
        with open("/proc/cpuinfo") as fh:
            flags = fh.read()
        return all(f in flags for f in (" avx2", " bmi2", " popcnt", " aes"))
    except OSError:
        return False
