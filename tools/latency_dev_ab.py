"""Device-pointer small-batch latency on the default and on a caller-owned stream: the A/B tool of the
second-stream matrix expansion (KYBER_HIP_LIBRARY selects the variant, tools/build_variant.sh)."""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
import kyber_amd
kem = kyber_amd.KyberHip(3, 0)
import os
SIZES = tuple(int(x) for x in os.environ.get("KY_LAT_SIZES", "1,256,4096").split(","))
for use_stream in ((False, True) if "KY_LAT_SIZES" not in os.environ else (False,)):
    st = torch.cuda.Stream() if use_stream else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        for n in SIZES:
            g = torch.Generator(device="cuda"); g.manual_seed(n)
            kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
            ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
            pk, sk = kem.keypair_dev(kc); ct, ss = kem.enc_dev(pk, ec); ss2 = torch.empty_like(ss)
            res = {}
            for name, fn in (("keypair", lambda: kem.keypair_dev(kc, pk, sk)), ("enc", lambda: kem.enc_dev(pk, ec, ct, ss)), ("dec", lambda: kem.dec_dev(ct, sk, ss2))):
                ts = []
                for _ in range(200):
                    st.synchronize(); t0 = time.perf_counter(); fn(); st.synchronize(); ts.append(time.perf_counter() - t0)
                res[name] = np.median(ts) * 1e6
            print("stream=%s n=%5d keypair %6.1f enc %6.1f dec %6.1f" % ("own" if use_stream else "null", n, res["keypair"], res["enc"], res["dec"]))
