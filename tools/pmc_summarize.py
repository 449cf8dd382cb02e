#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel name:
   python3 tools/pmc_summarize.py gpurun_out/pmc_*/ > profiles/rNN_pmc_summary.txt
   python3 tools/pmc_summarize.py --traffic-json gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_pmc_traffic.json

--traffic-json prices HBM bytes per launch the way MI355X_MICROARCH.md's HBM
section prescribes: FETCH_SIZE and WRITE_SIZE are KiB; on gfx950 FETCH_SIZE
reports half of a wide coalesced read stream, so it is doubled; WRITE_SIZE is
taken as is."""
import csv, glob, json, sys, collections, re
as_json = "--traffic-json" in sys.argv
note = "bench.py --steps 1 --warmup 1 --batch 524288 (one chunk of 2^19 instances per launch)"
for a in sys.argv:
    if a.startswith("--note="):
        note = a[len("--note="):]
sys.argv = [a for a in sys.argv if a != "--traffic-json" and not a.startswith("--note=")]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
            cnt[k][row["Counter_Name"]] += 1
if as_json:
    out = {}
    for k in sorted(agg):
        if not k.startswith("ky::") or "FETCH_SIZE" not in agg[k] or "WRITE_SIZE" not in agg[k]:
            continue
        f = agg[k]["FETCH_SIZE"] / cnt[k]["FETCH_SIZE"]
        w = agg[k]["WRITE_SIZE"] / cnt[k]["WRITE_SIZE"]
        out[k] = {"launches": cnt[k]["FETCH_SIZE"], "FETCH_SIZE_KiB_per_launch": f, "WRITE_SIZE_KiB_per_launch": w,
                  "hbm_bytes_per_launch": (2 * f + w) * 1024}
    import hashlib, os
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "kyber_amd", "libkyber_hip.so")
    print(json.dumps({"note": note + "; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
                      "lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None,  # bench.py drops figures of another build
                      "kernels": out}, indent=1))
    sys.exit(0)
for k in sorted(agg):
    if not k.startswith("ky::"):
        continue
    print(k)
    for c in sorted(agg[k]):
        print("   %-28s total %16.0f  launches %4d  per-launch %16.1f" % (c, agg[k][c], cnt[k][c], agg[k][c] / cnt[k][c]))
