import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
import oracle_lib, kyber_amd
o = oracle_lib.load_oracle()
for k in (3, 2, 4):
    kem = kyber_amd.KyberHip(k)
    n = 300
    rng = np.random.default_rng(k)
    kc = rng.integers(0, 256, (n, 64), dtype=np.uint8); ec = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    pk, sk = kem.keypair(kc); opk, osk = o.keypair(k, kc)
    print(k, 'pk', np.array_equal(pk, opk), 'sk', np.array_equal(sk, osk), flush=True)
    if not np.array_equal(pk, opk):
        d = np.argwhere(pk != opk); print('first pk diffs', d[:5], len(d))
    if not np.array_equal(sk, osk):
        d = np.argwhere(sk != osk); print('first sk diffs', d[:5], len(d))
    ct, ss = kem.enc(opk, ec); oct_, oss = o.enc(k, opk, ec)
    print(k, 'ct', np.array_equal(ct, oct_), 'ss', np.array_equal(ss, oss), flush=True)
    if not np.array_equal(ct, oct_):
        d = np.argwhere(ct != oct_); print('first ct diffs', d[:5], len(d))
    ss2 = kem.dec(oct_, osk)
    print(k, 'dec', np.array_equal(ss2, oss), flush=True)
