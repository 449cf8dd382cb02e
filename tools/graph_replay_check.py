"""kyber_batch_dec_dev_noalloc captured in a HIP graph (torch.cuda.CUDAGraph) and replayed six times, for every parameter set
and batch sizes on both sides of a multiple of 8, next to six eager calls back to back.  Found in round 3: with the tail
queue counter cleared by hipMemsetAsync, the memset NODE of the graph took effect on the first launch only and every later
replay returned a wrong secret for instance 0 when n was not a multiple of 8 (and, after a change of parameter set, read a
garbage counter); the counter is now cleared by a kernel (kyber_hip.hip: k_zero32)."""
import sys, torch
sys.path.insert(0, ".")
import kyber_amd
for k, n90 in ((3, 1), (2, 0), (4, 0), (2, 1), (4, 1), (3, 0)):
    kem = kyber_amd.KyberHip(k, n90)
    for n in (1, 2, 3, 4, 7, 8, 9, 16, 32, 33):
        g = torch.Generator(device="cuda"); g.manual_seed(n)
        kc = torch.randint(0, 256, (n, 64), dtype=torch.uint8, device="cuda", generator=g)
        ec = torch.randint(0, 256, (n, 32), dtype=torch.uint8, device="cuda", generator=g)
        pk, sk = kem.keypair_dev(kc); ct, ss = kem.enc_dev(pk, ec)
        out = torch.zeros_like(ss)
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            kem.dec_dev_noalloc(ct, sk, out)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                kem.dec_dev_noalloc(ct, sk, out)
        torch.cuda.synchronize()
        bad = []
        for r in range(6):
            out.zero_(); torch.cuda.synchronize()
            graph.replay(); torch.cuda.synchronize()
            bad.append([int(i) for i in (out != ss).any(dim=1).nonzero().flatten()])
        # eager calls back to back, no synchronisation in between
        out.zero_()
        for r in range(6):
            kem.dec_dev_noalloc(ct, sk, out)
        torch.cuda.synchronize()
        print(k, n90, n, "replays: wrong instances", bad, "| eager x6 ok", torch.equal(out, ss), flush=True)
