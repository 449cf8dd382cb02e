"""ctypes binding of libkyber_hip.so (see include/kyber_batch.h).

Mirrors the reference's three-function KEM interface (ref/api.h:16-18,39-41,
62-64; ref/kem.c:23-127) in batched form:

    kem = KyberHip(3)                    # Kyber768
    pk, sk = kem.keypair(coins)          # coins: uint8[n,64] = d||z per instance
    ct, ss = kem.enc(pk, coins)          # coins: uint8[n,32] = m0 per instance
    ss2    = kem.dec(ct, sk)

Host calls take/return numpy uint8 arrays; the *_dev calls take torch CUDA
uint8 tensors (device-resident, no PCIe traffic) and enqueue on torch's
current stream.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

PARAMETER_SETS = {
    # name: (k, nineties)                                   ref/params.h:11-31
    "kyber512": (2, 0), "kyber768": (3, 0), "kyber1024": (4, 0),
    "kyber512-90s": (2, 1), "kyber768-90s": (3, 1), "kyber1024-90s": (4, 1),
}


class KyberHipError(RuntimeError):
    pass


def library_path():
    """In-tree libkyber_hip.so; KYBER_HIP_LIBRARY names another build of the same
    library (A/B measurements, tools/ab_bench.sh)."""
    return os.environ.get("KYBER_HIP_LIBRARY") or os.path.join(_HERE, "libkyber_hip.so")


def sizes(k):
    """(pk, sk, ct) bytes of parameter set k -- ref/params.h:63-66."""
    du, dv = (11, 5) if k == 4 else (10, 4)
    return 384 * k + 32, 768 * k + 96, 32 * (du * k + dv)


def library():
    """Load libkyber_hip.so once; no fallback if it is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise KyberHipError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
    # torch ships its own copy of the HIP/HSA runtime.  Whichever copy a process
    # loads first must serve both: with the system runtime (this library's
    # dependency) loaded first, torch's own then finds no device.  So let torch,
    # where it is installed, load its runtime before the library is opened.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(path)
    u8p, i16p, sz, vp, ci = (ctypes.c_void_p,) * 2 + (ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int)
    sig = {
        "kyber_batch_keypair": [ci, ci, sz, u8p, u8p, u8p],
        "kyber_batch_enc": [ci, ci, sz, u8p, u8p, u8p, u8p],
        "kyber_batch_dec": [ci, ci, sz, u8p, u8p, u8p],
        "kyber_batch_keypair_mgpu": [ci, ci, sz, u8p, u8p, u8p, ci, vp],
        "kyber_batch_enc_mgpu": [ci, ci, sz, u8p, u8p, u8p, u8p, ci, vp],
        "kyber_batch_dec_mgpu": [ci, ci, sz, u8p, u8p, u8p, ci, vp],
        "kyber_batch_keypair_mgpu_dev": [ci, ci, ci, vp, vp, vp, vp, vp, vp],
        "kyber_batch_enc_mgpu_dev": [ci, ci, ci, vp, vp, vp, vp, vp, vp, vp],
        "kyber_batch_dec_mgpu_dev": [ci, ci, ci, vp, vp, vp, vp, vp, vp],
        "mlkem_batch_keypair": [ci, sz, u8p, u8p, u8p],
        "mlkem_batch_encaps": [ci, sz, u8p, u8p, u8p, u8p],
        "mlkem_batch_decaps": [ci, sz, u8p, u8p, u8p],
        "mlkem_batch_keypair_dev": [ci, sz, u8p, u8p, u8p, vp],
        "mlkem_batch_encaps_dev": [ci, sz, u8p, u8p, u8p, u8p, vp],
        "mlkem_batch_decaps_dev": [ci, sz, u8p, u8p, u8p, vp],
        "mlkem_batch_check_ek": [ci, sz, u8p, u8p],
        "mlkem_batch_check_dk": [ci, sz, u8p, u8p],
        "mlkem_batch_check_dev": [ci, sz, u8p, ci, u8p, vp],
        "kyber_batch_keypair_dev": [ci, ci, sz, u8p, u8p, u8p, vp],
        "kyber_batch_enc_dev": [ci, ci, sz, u8p, u8p, u8p, u8p, vp],
        "kyber_batch_dec_dev": [ci, ci, sz, u8p, u8p, u8p, vp],
        "kyber_batch_dec_dev_noalloc": [ci, ci, sz, u8p, u8p, u8p, vp],
        "kyber_batch_enc_shared_pk": [ci, ci, sz, u8p, u8p, u8p, u8p],
        "kyber_batch_dec_shared_sk": [ci, ci, sz, u8p, u8p, u8p],
        "kyber_batch_enc_shared_pk_dev": [ci, ci, sz, u8p, u8p, u8p, u8p, vp],
        "kyber_batch_dec_shared_sk_dev": [ci, ci, sz, u8p, u8p, u8p, vp],
        "kyber_hip_prepare_pk": [ci, ci, u8p, vp],
        "kyber_hip_prepare_sk": [ci, ci, u8p, vp],
        "kyber_hip_prepare_pk_dev": [ci, ci, u8p, vp],
        "kyber_hip_prepare_sk_dev": [ci, ci, u8p, vp],
        "kyber_hip_prepared_free": [vp],
        "kyber_batch_enc_prepared": [vp, sz, u8p, u8p, u8p],
        "kyber_batch_dec_prepared": [vp, sz, u8p, u8p],
        "kyber_batch_enc_prepared_dev": [vp, sz, u8p, u8p, u8p, vp],
        "kyber_batch_dec_prepared_dev": [vp, sz, u8p, u8p, vp],
        "kyber_batch_gen_matrix_dev": [ci, ci, sz, i16p, u8p, ci, vp],
        "kyber_batch_keccak_f1600_dev": [sz, vp, vp],
        "kyber_batch_hash_h_dev": [ci, ci, sz, ci, u8p, u8p, vp],
        "kyber_batch_hash_g_dev": [ci, sz, sz, u8p, u8p, vp],
        "kyber_batch_aes256ctr_dev": [sz, u8p, sz, u8p, vp, ctypes.c_uint32, vp],
        "kyber_batch_getnoise_dev": [ci, ci, sz, ci, ci, i16p, u8p, vp],
        "kyber_batch_poly_cbd_dev": [ci, sz, i16p, u8p, vp],
        "kyber_batch_poly_tobytes_dev": [sz, u8p, i16p, vp],
        "kyber_batch_poly_frombytes_dev": [sz, i16p, u8p, vp],
        "kyber_batch_poly_compress_dev": [ci, sz, u8p, i16p, vp],
        "kyber_batch_poly_decompress_dev": [ci, sz, i16p, u8p, vp],
        "kyber_batch_polyvec_basemul_acc_dev": [ci, sz, i16p, i16p, i16p, vp],
        "kyber_batch_poly_ntt_dev": [sz, i16p, vp],
        "kyber_batch_poly_invntt_dev": [sz, i16p, vp],
        "kyber_batch_poly_basemul_dev": [sz, i16p, i16p, i16p, vp],
        "kyber_batch_poly_mul_fused_dev": [sz, i16p, i16p, i16p, vp],
        "kyber_batch_poly_ntt_canon_dev": [sz, i16p, vp],
        "kyber_batch_poly_invntt_canon_dev": [sz, i16p, vp],
        "kyber_batch_poly_basemul_canon_dev": [sz, i16p, i16p, i16p, vp],
        "kyber_batch_poly_mul_fused_canon_dev": [sz, i16p, i16p, i16p, vp],
        "kyber_hip_set_chunk": [sz],
        "kyber_hip_set_tail_capacity": [sz],
        "kyber_hip_set_workspace_slots": [ci],
        "kyber_hip_profile_enable": [ci],
        "kyber_hip_profile_read": [ctypes.c_char_p, sz],
        "kyber_hip_release": [],
    }
    lib.kyber_hip_host_alloc.argtypes, lib.kyber_hip_host_alloc.restype = [sz], ctypes.c_void_p
    lib.kyber_hip_host_free.argtypes, lib.kyber_hip_host_free.restype = [ctypes.c_void_p], None
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = ctypes.c_int
    lib.kyber_hip_noalloc_max_n.argtypes, lib.kyber_hip_noalloc_max_n.restype = [], ctypes.c_size_t
    for name in ("kyber_hip_pk_bytes", "kyber_hip_sk_bytes", "kyber_hip_ct_bytes"):
        fn = getattr(lib, name)
        fn.argtypes = [ctypes.c_int]
        fn.restype = ctypes.c_size_t
    lib.kyber_hip_last_error.restype = ctypes.c_char_p
    _LIB = lib
    return lib


def _check(rc, what):
    if rc != 0:
        msg = library().kyber_hip_last_error().decode()
        names = {-1: "EINVAL", -2: "ENOSYS (parameter set not built)", -3: "EHIP"}
        raise KyberHipError("%s failed: %s %s" % (what, names.get(rc, rc), msg))


def _np_u8(a, shape, name):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    if a.shape != shape:
        raise ValueError("%s: expected shape %s, got %s" % (name, shape, a.shape))
    return a


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _torch_stream():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tptr(t, nbytes, name):
    if not t.is_cuda or not t.is_contiguous():
        raise ValueError("%s must be a contiguous CUDA tensor" % name)
    import torch
    if t.device.index != torch.cuda.current_device():
        # the library launches on the CURRENT device (workspace, locks, stream): a tensor of
        # another device would hand its kernels a foreign pointer
        raise ValueError("%s lives on cuda:%d but the current device is cuda:%d; use torch.cuda.device(...)"
                         % (name, t.device.index, torch.cuda.current_device()))
    if t.numel() * t.element_size() != nbytes:
        raise ValueError("%s: expected %d bytes, got %d" % (name, nbytes, t.numel() * t.element_size()))
    if t.data_ptr() % 16:
        raise ValueError("%s must be 16-byte aligned" % name)
    return ctypes.c_void_p(t.data_ptr())


class KyberHip:
    """One parameter set of the KEM (k = 2, 3, 4; nineties selects the -90s variant)."""

    def __init__(self, k=3, nineties=False):
        if k not in (2, 3, 4):
            raise ValueError("k must be 2, 3 or 4")
        self.k, self.nineties = k, int(bool(nineties))
        self.pk_bytes, self.sk_bytes, self.ct_bytes = sizes(k)
        self.lib = library()

    @classmethod
    def from_name(cls, name):
        k, n90 = PARAMETER_SETS[name.lower()]
        return cls(k, n90)

    # ---- host arrays ----
    def keypair(self, coins=None, n=None):
        if coins is not None:
            coins = np.ascontiguousarray(coins, dtype=np.uint8).reshape(-1, 64)
            n = coins.shape[0]
        elif n is None:
            raise ValueError("give coins or n")
        pk = np.empty((n, self.pk_bytes), np.uint8)
        sk = np.empty((n, self.sk_bytes), np.uint8)
        _check(self.lib.kyber_batch_keypair(self.k, self.nineties, n, _ptr(pk), _ptr(sk),
                                            _ptr(coins) if coins is not None else None), "kyber_batch_keypair")
        return pk, sk

    def enc(self, pk, coins=None):
        pk = np.ascontiguousarray(pk, dtype=np.uint8).reshape(-1, self.pk_bytes)
        n = pk.shape[0]
        if coins is not None:
            coins = _np_u8(np.asarray(coins).reshape(-1, 32), (n, 32), "coins")
        ct = np.empty((n, self.ct_bytes), np.uint8)
        ss = np.empty((n, 32), np.uint8)
        _check(self.lib.kyber_batch_enc(self.k, self.nineties, n, _ptr(ct), _ptr(ss), _ptr(pk),
                                        _ptr(coins) if coins is not None else None), "kyber_batch_enc")
        return ct, ss

    def dec(self, ct, sk):
        ct = np.ascontiguousarray(ct, dtype=np.uint8).reshape(-1, self.ct_bytes)
        n = ct.shape[0]
        sk = _np_u8(np.asarray(sk).reshape(-1, self.sk_bytes), (n, self.sk_bytes), "sk")
        ss = np.empty((n, 32), np.uint8)
        _check(self.lib.kyber_batch_dec(self.k, self.nineties, n, _ptr(ss), _ptr(ct), _ptr(sk)), "kyber_batch_dec")
        return ss

    # ---- several GPUs of one node (host arrays; contiguous shards, one host thread per device) ----
    @staticmethod
    def _devs(devices):
        devs = list(devices)
        return len(devs), (ctypes.c_int * len(devs))(*devs)

    def keypair_mgpu(self, coins, devices):
        coins = np.ascontiguousarray(coins, dtype=np.uint8).reshape(-1, 64)
        n = coins.shape[0]
        pk, sk = np.empty((n, self.pk_bytes), np.uint8), np.empty((n, self.sk_bytes), np.uint8)
        nd, arr = self._devs(devices)
        _check(self.lib.kyber_batch_keypair_mgpu(self.k, self.nineties, n, _ptr(pk), _ptr(sk), _ptr(coins), nd, arr),
               "kyber_batch_keypair_mgpu")
        return pk, sk

    def enc_mgpu(self, pk, coins, devices):
        pk = np.ascontiguousarray(pk, dtype=np.uint8).reshape(-1, self.pk_bytes)
        n = pk.shape[0]
        coins = _np_u8(np.asarray(coins).reshape(-1, 32), (n, 32), "coins")
        ct, ss = np.empty((n, self.ct_bytes), np.uint8), np.empty((n, 32), np.uint8)
        nd, arr = self._devs(devices)
        _check(self.lib.kyber_batch_enc_mgpu(self.k, self.nineties, n, _ptr(ct), _ptr(ss), _ptr(pk), _ptr(coins), nd, arr),
               "kyber_batch_enc_mgpu")
        return ct, ss

    def dec_mgpu(self, ct, sk, devices):
        ct = np.ascontiguousarray(ct, dtype=np.uint8).reshape(-1, self.ct_bytes)
        n = ct.shape[0]
        sk = _np_u8(np.asarray(sk).reshape(-1, self.sk_bytes), (n, self.sk_bytes), "sk")
        ss = np.empty((n, 32), np.uint8)
        nd, arr = self._devs(devices)
        _check(self.lib.kyber_batch_dec_mgpu(self.k, self.nineties, n, _ptr(ss), _ptr(ct), _ptr(sk), nd, arr),
               "kyber_batch_dec_mgpu")
        return ss

    # ---- one key, many operations (host arrays) ----
    def enc_shared_pk(self, pk, coins=None, n=None):
        """n encapsulations to the ONE public key pk; == enc(np.tile(pk, (n, 1)), coins)."""
        pk = _np_u8(np.asarray(pk).reshape(-1), (self.pk_bytes,), "pk")
        if coins is not None:
            coins = np.ascontiguousarray(coins, dtype=np.uint8).reshape(-1, 32)
            n = coins.shape[0]
        elif n is None:
            raise ValueError("give coins or n")
        ct = np.empty((n, self.ct_bytes), np.uint8)
        ss = np.empty((n, 32), np.uint8)
        _check(self.lib.kyber_batch_enc_shared_pk(self.k, self.nineties, n, _ptr(ct), _ptr(ss), _ptr(pk),
                                                  _ptr(coins) if coins is not None else None), "kyber_batch_enc_shared_pk")
        return ct, ss

    def dec_shared_sk(self, ct, sk):
        """Decapsulate every ciphertext with the ONE secret key sk."""
        ct = np.ascontiguousarray(ct, dtype=np.uint8).reshape(-1, self.ct_bytes)
        n = ct.shape[0]
        sk = _np_u8(np.asarray(sk).reshape(-1), (self.sk_bytes,), "sk")
        ss = np.empty((n, 32), np.uint8)
        _check(self.lib.kyber_batch_dec_shared_sk(self.k, self.nineties, n, _ptr(ss), _ptr(ct), _ptr(sk)),
               "kyber_batch_dec_shared_sk")
        return ss

    # ---- device tensors (torch, uint8, CUDA) ----
    def prepare_pk(self, pk):
        """-> PreparedKey for encapsulations to pk (numpy array or CUDA tensor)"""
        return PreparedKey(self, pk, False)

    def prepare_sk(self, sk):
        """-> PreparedKey for decapsulations with sk"""
        return PreparedKey(self, sk, True)

    def enc_shared_pk_dev(self, pk, coins, ct=None, ss=None):
        import torch
        n = coins.numel() // 32
        ct = torch.empty((n, self.ct_bytes), dtype=torch.uint8, device=pk.device) if ct is None else ct
        ss = torch.empty((n, 32), dtype=torch.uint8, device=pk.device) if ss is None else ss
        if n == 0:  # empty tensors have no data pointer; same result as the host path
            return ct, ss
        _check(self.lib.kyber_batch_enc_shared_pk_dev(self.k, self.nineties, n, _tptr(ct, n * self.ct_bytes, "ct"),
                                                      _tptr(ss, n * 32, "ss"), _tptr(pk, self.pk_bytes, "pk"),
                                                      _tptr(coins, n * 32, "coins"), _torch_stream()),
               "kyber_batch_enc_shared_pk_dev")
        return ct, ss

    def dec_shared_sk_dev(self, ct, sk, ss=None):
        import torch
        n = ct.numel() // self.ct_bytes
        ss = torch.empty((n, 32), dtype=torch.uint8, device=ct.device) if ss is None else ss
        if n == 0:
            return ss
        _check(self.lib.kyber_batch_dec_shared_sk_dev(self.k, self.nineties, n, _tptr(ss, n * 32, "ss"),
                                                      _tptr(ct, n * self.ct_bytes, "ct"), _tptr(sk, self.sk_bytes, "sk"),
                                                      _torch_stream()), "kyber_batch_dec_shared_sk_dev")
        return ss

    def keypair_dev(self, coins, pk=None, sk=None):
        import torch
        n = coins.numel() // 64
        pk = torch.empty((n, self.pk_bytes), dtype=torch.uint8, device=coins.device) if pk is None else pk
        sk = torch.empty((n, self.sk_bytes), dtype=torch.uint8, device=coins.device) if sk is None else sk
        if n == 0:
            return pk, sk
        _check(self.lib.kyber_batch_keypair_dev(self.k, self.nineties, n, _tptr(pk, n * self.pk_bytes, "pk"),
                                                _tptr(sk, n * self.sk_bytes, "sk"), _tptr(coins, n * 64, "coins"),
                                                _torch_stream()), "kyber_batch_keypair_dev")
        return pk, sk

    def enc_dev(self, pk, coins, ct=None, ss=None):
        import torch
        n = coins.numel() // 32
        ct = torch.empty((n, self.ct_bytes), dtype=torch.uint8, device=pk.device) if ct is None else ct
        ss = torch.empty((n, 32), dtype=torch.uint8, device=pk.device) if ss is None else ss
        if n == 0:
            return ct, ss
        _check(self.lib.kyber_batch_enc_dev(self.k, self.nineties, n, _tptr(ct, n * self.ct_bytes, "ct"),
                                            _tptr(ss, n * 32, "ss"), _tptr(pk, n * self.pk_bytes, "pk"),
                                            _tptr(coins, n * 32, "coins"), _torch_stream()), "kyber_batch_enc_dev")
        return ct, ss

    def dec_dev(self, ct, sk, ss=None):
        import torch
        n = ct.numel() // self.ct_bytes
        ss = torch.empty((n, 32), dtype=torch.uint8, device=ct.device) if ss is None else ss
        if n == 0:
            return ss
        _check(self.lib.kyber_batch_dec_dev(self.k, self.nineties, n, _tptr(ss, n * 32, "ss"),
                                            _tptr(ct, n * self.ct_bytes, "ct"), _tptr(sk, n * self.sk_bytes, "sk"),
                                            _torch_stream()), "kyber_batch_dec_dev")
        return ss

    def dec_dev_noalloc(self, ct, sk, ss):
        """kyber_batch_dec_dev_noalloc: as dec_dev, but the library's workspace is its
        fixed device array (no allocation anywhere; ss must be given; n <= noalloc_max_n())."""
        n = ct.numel() // self.ct_bytes
        if n == 0:
            return ss
        _check(self.lib.kyber_batch_dec_dev_noalloc(self.k, self.nineties, n, _tptr(ss, n * 32, "ss"),
                                                    _tptr(ct, n * self.ct_bytes, "ct"), _tptr(sk, n * self.sk_bytes, "sk"),
                                                    _torch_stream()), "kyber_batch_dec_dev_noalloc")
        return ss

    def gen_matrix_dev(self, seeds, transposed):
        """seeds: uint8[n,32] CUDA tensor -> int16[n, k*k, 256] (ref/indcpa.c:164-192)."""
        import torch
        n = seeds.numel() // 32
        mat = torch.empty((n, self.k * self.k, 256), dtype=torch.int16, device=seeds.device)
        if n == 0:
            return mat
        _check(self.lib.kyber_batch_gen_matrix_dev(self.k, self.nineties, n, _tptr(mat, mat.numel() * 2, "mat"),
                                                   _tptr(seeds, n * 32, "seeds"), int(bool(transposed)),
                                                   _torch_stream()), "kyber_batch_gen_matrix_dev")
        return mat


class PreparedKey:
    """One public or secret key with its expanded matrix and H(pk) resident on the current device
    (kyber_hip_prepare_pk / _sk): the shared-key calls without their per-call set-up.  Use as a
    context manager or call close(); a secret key's device copy is wiped on release."""

    def __init__(self, kem, key, secret):
        self.kem, self.secret = kem, bool(secret)
        want = kem.sk_bytes if secret else kem.pk_bytes
        h = ctypes.c_void_p()
        name = "kyber_hip_prepare_%s" % ("sk" if secret else "pk")
        if hasattr(key, "is_cuda"):
            if key.numel() != want:
                raise ValueError("key: expected %d bytes" % want)
            rc = getattr(kem.lib, name + "_dev")(kem.k, kem.nineties, _tptr(key, want, "key"), ctypes.byref(h))
        else:
            key = _np_u8(np.asarray(key).reshape(-1), (want,), "key")
            rc = getattr(kem.lib, name)(kem.k, kem.nineties, _ptr(key), ctypes.byref(h))
        _check(rc, name)
        self._h = h

    def close(self):
        if self._h is not None:
            h, self._h = self._h, None
            _check(self.kem.lib.kyber_hip_prepared_free(h), "kyber_hip_prepared_free")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enc(self, coins=None, n=None):
        """n encapsulations to the prepared public key (host arrays); == kem.enc_shared_pk(pk, coins)."""
        if coins is not None:
            coins = np.ascontiguousarray(coins, dtype=np.uint8).reshape(-1, 32)
            n = coins.shape[0]
        elif n is None:
            raise ValueError("give coins or n")
        ct, ss = np.empty((n, self.kem.ct_bytes), np.uint8), np.empty((n, 32), np.uint8)
        _check(self.kem.lib.kyber_batch_enc_prepared(self._h, n, _ptr(ct), _ptr(ss), _ptr(coins) if coins is not None else None),
               "kyber_batch_enc_prepared")
        return ct, ss

    def dec(self, ct):
        ct = np.ascontiguousarray(ct, dtype=np.uint8).reshape(-1, self.kem.ct_bytes)
        ss = np.empty((ct.shape[0], 32), np.uint8)
        _check(self.kem.lib.kyber_batch_dec_prepared(self._h, ct.shape[0], _ptr(ss), _ptr(ct)), "kyber_batch_dec_prepared")
        return ss

    def enc_dev(self, coins, ct=None, ss=None):
        import torch
        n = coins.numel() // 32
        ct = torch.empty((n, self.kem.ct_bytes), dtype=torch.uint8, device=coins.device) if ct is None else ct
        ss = torch.empty((n, 32), dtype=torch.uint8, device=coins.device) if ss is None else ss
        if n:
            _check(self.kem.lib.kyber_batch_enc_prepared_dev(self._h, n, _tptr(ct, n * self.kem.ct_bytes, "ct"), _tptr(ss, n * 32, "ss"),
                                                             _tptr(coins, n * 32, "coins"), _torch_stream()), "kyber_batch_enc_prepared_dev")
        return ct, ss

    def dec_dev(self, ct, ss=None):
        import torch
        n = ct.numel() // self.kem.ct_bytes
        ss = torch.empty((n, 32), dtype=torch.uint8, device=ct.device) if ss is None else ss
        if n:
            _check(self.kem.lib.kyber_batch_dec_prepared_dev(self._h, n, _tptr(ss, n * 32, "ss"), _tptr(ct, n * self.kem.ct_bytes, "ct"),
                                                             _torch_stream()), "kyber_batch_dec_prepared_dev")
        return ss


class MlKemHip:
    """ML-KEM (FIPS 203) mode, include/mlkem_batch.h: the *_internal algorithms, batched.
    Not in the reference; parity unpinned (see the header).  k = 2, 3, 4."""

    def __init__(self, k=3):
        if k not in (2, 3, 4):
            raise ValueError("k must be 2, 3 or 4")
        self.k = k
        self.ek_bytes, self.dk_bytes, self.ct_bytes = sizes(k)
        self.lib = library()

    def keypair(self, coins):
        coins = np.ascontiguousarray(coins, dtype=np.uint8).reshape(-1, 64)
        n = coins.shape[0]
        ek, dk = np.empty((n, self.ek_bytes), np.uint8), np.empty((n, self.dk_bytes), np.uint8)
        _check(self.lib.mlkem_batch_keypair(self.k, n, _ptr(ek), _ptr(dk), _ptr(coins)), "mlkem_batch_keypair")
        return ek, dk

    def encaps(self, ek, coins):
        ek = np.ascontiguousarray(ek, dtype=np.uint8).reshape(-1, self.ek_bytes)
        n = ek.shape[0]
        coins = _np_u8(np.asarray(coins).reshape(-1, 32), (n, 32), "coins")
        ct, ss = np.empty((n, self.ct_bytes), np.uint8), np.empty((n, 32), np.uint8)
        _check(self.lib.mlkem_batch_encaps(self.k, n, _ptr(ct), _ptr(ss), _ptr(ek), _ptr(coins)), "mlkem_batch_encaps")
        return ct, ss

    def decaps(self, ct, dk):
        ct = np.ascontiguousarray(ct, dtype=np.uint8).reshape(-1, self.ct_bytes)
        n = ct.shape[0]
        dk = _np_u8(np.asarray(dk).reshape(-1, self.dk_bytes), (n, self.dk_bytes), "dk")
        ss = np.empty((n, 32), np.uint8)
        _check(self.lib.mlkem_batch_decaps(self.k, n, _ptr(ss), _ptr(ct), _ptr(dk)), "mlkem_batch_decaps")
        return ss

    def check_ek(self, ek):
        ek = np.ascontiguousarray(ek, dtype=np.uint8).reshape(-1, self.ek_bytes)
        flags = np.empty(ek.shape[0], np.uint32)
        _check(self.lib.mlkem_batch_check_ek(self.k, ek.shape[0], _ptr(ek), _ptr(flags)), "mlkem_batch_check_ek")
        return flags

    def check_dk(self, dk):
        dk = np.ascontiguousarray(dk, dtype=np.uint8).reshape(-1, self.dk_bytes)
        flags = np.empty(dk.shape[0], np.uint32)
        _check(self.lib.mlkem_batch_check_dk(self.k, dk.shape[0], _ptr(dk), _ptr(flags)), "mlkem_batch_check_dk")
        return flags

    def keypair_dev(self, coins):
        import torch
        n = coins.numel() // 64
        ek = torch.empty((n, self.ek_bytes), dtype=torch.uint8, device=coins.device)
        dk = torch.empty((n, self.dk_bytes), dtype=torch.uint8, device=coins.device)
        if n:
            _check(self.lib.mlkem_batch_keypair_dev(self.k, n, _tptr(ek, n * self.ek_bytes, "ek"), _tptr(dk, n * self.dk_bytes, "dk"),
                                                    _tptr(coins, n * 64, "coins"), _torch_stream()), "mlkem_batch_keypair_dev")
        return ek, dk

    def encaps_dev(self, ek, coins):
        import torch
        n = coins.numel() // 32
        ct = torch.empty((n, self.ct_bytes), dtype=torch.uint8, device=ek.device)
        ss = torch.empty((n, 32), dtype=torch.uint8, device=ek.device)
        if n:
            _check(self.lib.mlkem_batch_encaps_dev(self.k, n, _tptr(ct, n * self.ct_bytes, "ct"), _tptr(ss, n * 32, "ss"),
                                                   _tptr(ek, n * self.ek_bytes, "ek"), _tptr(coins, n * 32, "coins"), _torch_stream()),
                   "mlkem_batch_encaps_dev")
        return ct, ss

    def decaps_dev(self, ct, dk):
        import torch
        n = ct.numel() // self.ct_bytes
        ss = torch.empty((n, 32), dtype=torch.uint8, device=ct.device)
        if n:
            _check(self.lib.mlkem_batch_decaps_dev(self.k, n, _tptr(ss, n * 32, "ss"), _tptr(ct, n * self.ct_bytes, "ct"),
                                                   _tptr(dk, n * self.dk_bytes, "dk"), _torch_stream()), "mlkem_batch_decaps_dev")
        return ss


def profile_enable(on):
    """Bracket every kernel launch with HIP events (measurement aid)."""
    _check(library().kyber_hip_profile_enable(int(bool(on))), "kyber_hip_profile_enable")


def profile_read():
    """-> {kernel name: (launches, total_ms)} since profile_enable(True)."""
    buf = ctypes.create_string_buffer(8192)
    _check(library().kyber_hip_profile_read(buf, len(buf)), "kyber_hip_profile_read")
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


def noalloc_max_n():
    return int(library().kyber_hip_noalloc_max_n())


def set_chunk(instances):
    _check(library().kyber_hip_set_chunk(int(instances)), "kyber_hip_set_chunk")


def set_workspace_slots(slots):
    """Workspaces per device (1..8): with more than one, batches of up to 65 536 instances on different
    streams run side by side on the device (include/kyber_batch.h: kyber_hip_set_workspace_slots)."""
    _check(library().kyber_hip_set_workspace_slots(int(slots)), "kyber_hip_set_workspace_slots")


class _Pinned:
    def __init__(self, nbytes):
        self.lib = library()
        self.ptr = self.lib.kyber_hip_host_alloc(nbytes)
        if not self.ptr:
            raise KyberHipError("kyber_hip_host_alloc(%d) failed" % nbytes)

    def __del__(self):
        if getattr(self, "ptr", None):
            self.lib.kyber_hip_host_free(self.ptr)
            self.ptr = None


def pinned_empty(shape):
    """uint8 numpy array in page-locked host memory (kyber_hip_host_alloc); the
    host-pointer calls reach full PCIe rate on such arrays.  The memory is
    released when the array (and every view of it) is gone."""
    shape = tuple(int(x) for x in (shape if isinstance(shape, (tuple, list)) else (shape,)))
    n = int(np.prod(shape)) if shape else 1
    owner = _Pinned(max(n, 1))
    buf = (ctypes.c_uint8 * max(n, 1)).from_address(owner.ptr)
    buf._pin = owner  # the array's base object keeps the allocation alive
    return np.frombuffer(buf, dtype=np.uint8, count=n).reshape(shape)


def set_tail_capacity(records=None):
    """Bound the matrix sampler's tail queue (None = default); output never changes."""
    v = (1 << 64) - 1 if records is None else int(records)
    _check(library().kyber_hip_set_tail_capacity(v), "kyber_hip_set_tail_capacity")


# ---- batched polynomial kernels on torch int16 tensors [npoly, 256] ----
def _poly_call(name, *tensors):
    lib = library()
    npoly = tensors[0].numel() // 256
    args = [_tptr(t, npoly * 512, "poly") for t in tensors]
    _check(getattr(lib, name)(npoly, *args, _torch_stream()), name)


def poly_ntt_(r):
    _poly_call("kyber_batch_poly_ntt_dev", r)
    return r


def poly_invntt_(r):
    _poly_call("kyber_batch_poly_invntt_dev", r)
    return r


def poly_basemul(r, a, b):
    _poly_call("kyber_batch_poly_basemul_dev", r, a, b)
    return r


def poly_mul_fused(r, a, b):
    _poly_call("kyber_batch_poly_mul_fused_dev", r, a, b)
    return r


# canonical-output forms (coefficients in [0, q)): the fast path, see include/kyber_batch.h
def poly_ntt_canon_(r):
    _poly_call("kyber_batch_poly_ntt_canon_dev", r)
    return r


def poly_invntt_canon_(r):
    _poly_call("kyber_batch_poly_invntt_canon_dev", r)
    return r


def poly_basemul_canon(r, a, b):
    _poly_call("kyber_batch_poly_basemul_canon_dev", r, a, b)
    return r


def poly_mul_fused_canon(r, a, b):
    _poly_call("kyber_batch_poly_mul_fused_canon_dev", r, a, b)
    return r


# ---- single-function stages (include/kyber_batch.h), torch tensors on the GPU ----
def getnoise(k, nineties, seeds, nnonce, n_eta1):
    """seeds uint8[n,32] -> int16[n, nnonce, 256]: poly_getnoise for nonces 0..nnonce-1 (ref/poly.c:208-232)."""
    import torch
    n = seeds.numel() // 32
    r = torch.empty((n, nnonce, 256), dtype=torch.int16, device=seeds.device)
    _check(library().kyber_batch_getnoise_dev(k, int(nineties), n, nnonce, n_eta1, _tptr(r, r.numel() * 2, "r"),
                                              _tptr(seeds, n * 32, "seeds"), _torch_stream()), "kyber_batch_getnoise_dev")
    return r


def poly_cbd(eta, buf):
    """buf uint8[npoly, 64*eta] -> int16[npoly,256] (ref/cbd.c:58-107)."""
    import torch
    npoly = buf.numel() // (64 * eta)
    r = torch.empty((npoly, 256), dtype=torch.int16, device=buf.device)
    _check(library().kyber_batch_poly_cbd_dev(eta, npoly, _tptr(r, npoly * 512, "r"), _tptr(buf, npoly * 64 * eta, "buf"),
                                              _torch_stream()), "kyber_batch_poly_cbd_dev")
    return r


def _codec(name, d, src, out_shape, out_dtype, in_bytes, out_bytes):
    import torch
    npoly = src.shape[0]
    out = torch.empty((npoly,) + out_shape, dtype=out_dtype, device=src.device)
    fn = getattr(library(), name)
    args = ([d] if d is not None else []) + [npoly, _tptr(out, npoly * out_bytes, "out"), _tptr(src, npoly * in_bytes, "in"), _torch_stream()]
    _check(fn(*args), name)
    return out


def poly_tobytes(a):
    import torch
    return _codec("kyber_batch_poly_tobytes_dev", None, a, (384,), torch.uint8, 512, 384)


def poly_frombytes(b):
    import torch
    return _codec("kyber_batch_poly_frombytes_dev", None, b, (256,), torch.int16, 384, 512)


def poly_compress(d, a):
    import torch
    return _codec("kyber_batch_poly_compress_dev", d, a, (32 * d,), torch.uint8, 512, 32 * d)


def poly_decompress(d, b):
    import torch
    return _codec("kyber_batch_poly_decompress_dev", d, b, (256,), torch.int16, 32 * d, 512)


def polyvec_basemul_acc(k, a, b):
    """a, b int16[nvec, k, 256] -> int16[nvec, 256], plain (non-Montgomery) residues; see the header."""
    import torch
    nvec = a.shape[0]
    r = torch.empty((nvec, 256), dtype=torch.int16, device=a.device)
    _check(library().kyber_batch_polyvec_basemul_acc_dev(k, nvec, _tptr(r, nvec * 512, "r"), _tptr(a, nvec * k * 512, "a"),
                                                         _tptr(b, nvec * k * 512, "b"), _torch_stream()),
           "kyber_batch_polyvec_basemul_acc_dev")
    return r


def keccak_f1600_(states):
    """int64/uint64 tensor [n,25] on the GPU, permuted in place (ref/fips202.c:82-344)."""
    n = states.numel() // 25
    _check(library().kyber_batch_keccak_f1600_dev(n, ctypes.c_void_p(states.data_ptr()), _torch_stream()), "kyber_batch_keccak_f1600_dev")
    return states


def aes256ctr(keys, nonces, nblocks, first_block=0):
    """uint8[n,32] keys, int16[n] nonces (the two leading nonce bytes, little-endian) -> uint8[n, nblocks, 16]:
    the keystream of the 90s sets' XOF / PRF from block `first_block` on."""
    import torch
    n = keys.shape[0]
    out = torch.empty((n, nblocks, 16), dtype=torch.uint8, device=keys.device)
    _check(library().kyber_batch_aes256ctr_dev(n, _tptr(out, n * nblocks * 16, "out"), nblocks, _tptr(keys, n * 32, "keys"),
                                               _tptr(nonces, n * 2, "nonces"), first_block, _torch_stream()),
           "kyber_batch_aes256ctr_dev")
    return out


def hash_h(k, nineties, data, of_ct):
    """uint8[n, PKB or CTB] -> uint8[n,32]: sha3_256 / sha256 as the FO kernels compute it."""
    import torch
    n = data.shape[0]
    out = torch.empty((n, 32), dtype=torch.uint8, device=data.device)
    _check(library().kyber_batch_hash_h_dev(k, int(nineties), n, int(bool(of_ct)), _tptr(out, n * 32, "out"),
                                            _tptr(data, data.numel(), "in"), _torch_stream()), "kyber_batch_hash_h_dev")
    return out


def hash_g(nineties, data):
    """uint8[n, 32 or 64] -> uint8[n,64]: sha3_512 / sha512 as the FO kernels compute it."""
    import torch
    n, inlen = data.shape
    out = torch.empty((n, 64), dtype=torch.uint8, device=data.device)
    _check(library().kyber_batch_hash_g_dev(int(nineties), n, inlen, _tptr(out, n * 64, "out"), _tptr(data, n * inlen, "in"), _torch_stream()),
           "kyber_batch_hash_g_dev")
    return out
