"""The reference's KEX test (ref/test_kex.c:8-67) as a batch scenario with
explicit coins, runnable over any implementation that offers keypair() and the
six kex functions (oracle, reference shim, GPU library)."""
import hashlib

import numpy as np


def coins_for(n, seed):
    rng = np.random.default_rng(seed)
    return {name: rng.integers(0, 256, (n, width), dtype=np.uint8)
            for name, width in (("kp_b", 64), ("kp_a", 64), ("uake_a", 96), ("uake_b", 32), ("ake_a", 96), ("ake_b", 64))}


def run(keypair, kex, c):
    """-> dict of every byte string the scenario produces."""
    out = {}
    out["pkb"], out["skb"] = keypair(c["kp_b"])      # static key of Bob      test_kex.c:29
    out["pka"], out["ska"] = keypair(c["kp_a"])      # static key of Alice    test_kex.c:30
    # unilaterally authenticated                                              test_kex.c:35-37
    out["uake_senda"], out["uake_tk"], out["uake_eska"] = kex.uake_initA(out["pkb"], c["uake_a"])
    out["uake_sendb"], out["uake_kb"] = kex.uake_sharedB(out["uake_senda"], out["skb"], c["uake_b"])
    out["uake_ka"] = kex.uake_sharedA(out["uake_sendb"], out["uake_tk"], out["uake_eska"])
    # mutually authenticated                                                  test_kex.c:47-49
    out["ake_senda"], out["ake_tk"], out["ake_eska"] = kex.ake_initA(out["pkb"], c["ake_a"])
    out["ake_sendb"], out["ake_kb"] = kex.ake_sharedB(out["ake_senda"], out["skb"], out["pka"], c["ake_b"])
    out["ake_ka"] = kex.ake_sharedA(out["ake_sendb"], out["ake_tk"], out["ake_eska"], out["ska"])
    return out


def check_agreement(out):
    """the reference's own pass criteria (test_kex.c:39-43, :51-55)"""
    assert np.array_equal(out["uake_ka"], out["uake_kb"]), "Error in UAKE"
    assert out["uake_ka"].any(axis=1).all(), "UAKE produces zero key"
    assert np.array_equal(out["ake_ka"], out["ake_kb"]), "Error in AKE"
    assert out["ake_ka"].any(axis=1).all(), "AKE produces zero key"


def digest(out):
    h = hashlib.sha256()
    for name in sorted(out):
        h.update(name.encode())
        h.update(np.ascontiguousarray(out[name], np.uint8).tobytes())
    return h.hexdigest()
