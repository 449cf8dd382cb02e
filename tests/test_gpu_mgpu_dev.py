"""GPU: kyber_batch_*_mgpu_dev -- shards that already live in device memory, one entry per device.
A one-GPU box names device 0 twice (allowed: its shards queue behind each other); results must be
those of the plain per-device calls and of the oracle (ref/kem.c:23-127, instance by instance)."""
import ctypes

import numpy as np
import pytest
import torch

import kyber_amd

pytestmark = pytest.mark.gpu


def _ptrs(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


@pytest.mark.parametrize("k,n90", [(3, 0), (4, 0), (2, 1)])
def test_mgpu_dev_two_shards_on_one_device(oracle, k, n90):
    kem = kyber_amd.KyberHip(k, n90)
    lib = kem.lib
    rng = np.random.default_rng(40 + k)
    ns = [37, 300]
    devs = (ctypes.c_int * 2)(0, 0)
    nn = (ctypes.c_size_t * 2)(*ns)
    kc = [torch.from_numpy(rng.integers(0, 256, (n, 64), dtype=np.uint8)).cuda() for n in ns]
    ec = [torch.from_numpy(rng.integers(0, 256, (n, 32), dtype=np.uint8)).cuda() for n in ns]
    mk = lambda w: [torch.empty((n, w), dtype=torch.uint8, device="cuda") for n in ns]
    pk, sk, ct, ss, ss2 = mk(kem.pk_bytes), mk(kem.sk_bytes), mk(kem.ct_bytes), mk(32), mk(32)
    s1 = torch.cuda.Stream()
    streams = (ctypes.c_void_p * 2)(None, s1.cuda_stream)  # shard 0 on the default stream, shard 1 on its own
    assert lib.kyber_batch_keypair_mgpu_dev(k, n90, 2, devs, nn, _ptrs(pk), _ptrs(sk), _ptrs(kc), streams) == 0
    assert lib.kyber_batch_enc_mgpu_dev(k, n90, 2, devs, nn, _ptrs(ct), _ptrs(ss), _ptrs(pk), _ptrs(ec), streams) == 0
    assert lib.kyber_batch_dec_mgpu_dev(k, n90, 2, devs, nn, _ptrs(ss2), _ptrs(ct), _ptrs(sk), None) == 0
    torch.cuda.synchronize()
    for g in range(2):
        opk, osk = oracle.keypair(k, kc[g].cpu().numpy(), n90)
        oct_, oss = oracle.enc(k, opk, ec[g].cpu().numpy(), n90)
        assert np.array_equal(pk[g].cpu().numpy(), opk) and np.array_equal(sk[g].cpu().numpy(), osk)
        assert np.array_equal(ct[g].cpu().numpy(), oct_) and np.array_equal(ss[g].cpu().numpy(), oss)
        assert np.array_equal(ss2[g].cpu().numpy(), oss)


def test_mgpu_dev_refuses_bad_arguments():
    lib = kyber_amd.library()
    devs = (ctypes.c_int * 1)(99)
    nn = (ctypes.c_size_t * 1)(1)
    t = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    p = _ptrs([t])
    assert lib.kyber_batch_dec_mgpu_dev(3, 0, 1, devs, nn, p, p, p, None) == -1   # no such device
    assert b"not present" in lib.kyber_hip_last_error()
    assert lib.kyber_batch_dec_mgpu_dev(3, 0, 0, devs, nn, p, p, p, None) == -1   # ndev < 1
    assert lib.kyber_batch_enc_mgpu_dev(3, 0, 1, None, nn, p, p, p, p, None) == -1  # devices must be given
