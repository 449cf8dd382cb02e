"""Batched key exchange: the reference's kex.h interface (ref/kex.h:7-29,
ref/kex.c:6-54) over dense numpy arrays, one row per protocol instance.
Every method is one call into libkyber_hip.so (kyber_batch_kex_*)."""
import ctypes

import numpy as np

from . import lib as _lib

SHARED_B = 1  # KYBER_KEX_SHARED_B: B's static key is one key for every instance


def _bind(lib):
    if getattr(lib, "_kex_bound", False):
        return
    ci, sz, p = ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p
    sig = {
        "kyber_batch_kex_uake_initA": [ci, ci, sz, p, p, p, p, p, ci],
        "kyber_batch_kex_ake_initA": [ci, ci, sz, p, p, p, p, p, ci],
        "kyber_batch_kex_uake_sharedB": [ci, ci, sz, p, p, p, p, p, ci],
        "kyber_batch_kex_uake_sharedA": [ci, ci, sz, p, p, p, p],
        "kyber_batch_kex_ake_sharedB": [ci, ci, sz, p, p, p, p, p, p, ci],
        "kyber_batch_kex_ake_sharedA": [ci, ci, sz, p, p, p, p, p],
        "kyber_batch_kex_uake_initA_dev": [ci, ci, sz, p, p, p, p, p, ci, p],
        "kyber_batch_kex_ake_initA_dev": [ci, ci, sz, p, p, p, p, p, ci, p],
        "kyber_batch_kex_uake_sharedB_dev": [ci, ci, sz, p, p, p, p, p, ci, p],
        "kyber_batch_kex_uake_sharedA_dev": [ci, ci, sz, p, p, p, p, p],
        "kyber_batch_kex_ake_sharedB_dev": [ci, ci, sz, p, p, p, p, p, p, ci, p],
        "kyber_batch_kex_ake_sharedA_dev": [ci, ci, sz, p, p, p, p, p, p],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = args, ci
    lib._kex_bound = True


class Kex:
    """kex_uake_* / kex_ake_* for one parameter set.  Sizes follow ref/kex.h:7-12."""

    def __init__(self, k=3, nineties=False):
        self.k, self.n90 = k, int(bool(nineties))
        self.pkb, self.skb, self.ctb = _lib.sizes(k)
        self.lib = _lib.library()
        _bind(self.lib)
        self.UAKE_SENDABYTES = self.AKE_SENDABYTES = self.pkb + self.ctb
        self.UAKE_SENDBBYTES = self.ctb
        self.AKE_SENDBBYTES = 2 * self.ctb
        self.SSBYTES = 32

    # -- helpers
    @staticmethod
    def _in(a, n, width, name):
        a = np.ascontiguousarray(a, dtype=np.uint8)
        if a.size != n * width:
            raise ValueError("%s: expected %d x %d bytes, got %d" % (name, n, width, a.size))
        return a

    @staticmethod
    def _p(a):
        return None if a is None else a.ctypes.data_as(ctypes.c_void_p)

    def _call(self, name, n, *args):
        _lib._check(getattr(self.lib, name)(self.k, self.n90, n, *args), name)

    def _initA(self, name, pkb, coins, n, shared_b):
        if coins is not None:
            coins = np.ascontiguousarray(coins, dtype=np.uint8).reshape(-1, 96)
            n = coins.shape[0]
        elif n is None:
            raise ValueError("give coins or n")
        pkb = self._in(pkb, 1 if shared_b else n, self.pkb, "pkb")
        send, tk, sk = (np.empty((n, w), np.uint8) for w in (self.pkb + self.ctb, 32, self.skb))
        self._call(name, n, self._p(send), self._p(tk), self._p(sk), self._p(pkb), self._p(coins), SHARED_B if shared_b else 0)
        return send, tk, sk

    # -- ref/kex.c:6-10, :30-34
    def uake_initA(self, pkb, coins=None, n=None, shared_b=False):
        return self._initA("kyber_batch_kex_uake_initA", pkb, coins, n, shared_b)

    def ake_initA(self, pkb, coins=None, n=None, shared_b=False):
        return self._initA("kyber_batch_kex_ake_initA", pkb, coins, n, shared_b)

    # -- ref/kex.c:12-18
    def uake_sharedB(self, recv, skb, coins=None, shared_b=False):
        recv = np.ascontiguousarray(recv, dtype=np.uint8).reshape(-1, self.pkb + self.ctb)
        n = recv.shape[0]
        skb = self._in(skb, 1 if shared_b else n, self.skb, "skb")
        coins = None if coins is None else self._in(coins, n, 32, "coins")
        send, key = np.empty((n, self.ctb), np.uint8), np.empty((n, 32), np.uint8)
        self._call("kyber_batch_kex_uake_sharedB", n, self._p(send), self._p(key), self._p(recv), self._p(skb), self._p(coins),
                   SHARED_B if shared_b else 0)
        return send, key

    # -- ref/kex.c:20-27
    def uake_sharedA(self, recv, tk, sk):
        recv = np.ascontiguousarray(recv, dtype=np.uint8).reshape(-1, self.ctb)
        n = recv.shape[0]
        key = np.empty((n, 32), np.uint8)
        self._call("kyber_batch_kex_uake_sharedA", n, self._p(key), self._p(recv), self._p(self._in(tk, n, 32, "tk")),
                   self._p(self._in(sk, n, self.skb, "sk")))
        return key

    # -- ref/kex.c:36-43
    def ake_sharedB(self, recv, skb, pka, coins=None, shared_b=False):
        recv = np.ascontiguousarray(recv, dtype=np.uint8).reshape(-1, self.pkb + self.ctb)
        n = recv.shape[0]
        skb = self._in(skb, 1 if shared_b else n, self.skb, "skb")
        pka = self._in(pka, n, self.pkb, "pka")
        coins = None if coins is None else self._in(coins, n, 64, "coins")
        send, key = np.empty((n, 2 * self.ctb), np.uint8), np.empty((n, 32), np.uint8)
        self._call("kyber_batch_kex_ake_sharedB", n, self._p(send), self._p(key), self._p(recv), self._p(skb), self._p(pka),
                   self._p(coins), SHARED_B if shared_b else 0)
        return send, key

    # -- ref/kex.c:45-54
    def ake_sharedA(self, recv, tk, sk, ska):
        recv = np.ascontiguousarray(recv, dtype=np.uint8).reshape(-1, 2 * self.ctb)
        n = recv.shape[0]
        key = np.empty((n, 32), np.uint8)
        self._call("kyber_batch_kex_ake_sharedA", n, self._p(key), self._p(recv), self._p(self._in(tk, n, 32, "tk")),
                   self._p(self._in(sk, n, self.skb, "sk")), self._p(self._in(ska, n, self.skb, "ska")))
        return key
