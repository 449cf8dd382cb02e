#!/usr/bin/env python3
"""tools/round_profiles_collect.py <run-prefix> <profile-prefix> -- copy what tools/round_profiles.sh left under
gpurun_out/ (scratch) into profiles/ (tracked): bench lines, rocprofv3 kernel stats, counter summaries.
   python3 tools/round_profiles_collect.py r02b r02"""
import glob
import os
import shutil
import subprocess
import sys

run, out = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
for f in glob.glob(os.path.join(G, run + "_bench_*.json")):
    shutil.copy(f, os.path.join(P, os.path.basename(f).replace(run + "_", out + "_")))
for f in glob.glob(os.path.join(G, run + "_smi_during_bench.txt")):
    shutil.copy(f, os.path.join(P, out + "_smi_during_bench.txt"))
for tag in ("kem", "ntt", "kem90s"):
    for f in glob.glob(os.path.join(G, "%s_prof_%s" % (run, tag), "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(P, "%s_%s_kernel_stats.csv" % (out, tag)))
summ = os.path.join(ROOT, "tools", "pmc_summarize.py")
for tag, name in (("", "kem"), ("_ntt", "ntt")):
    sq = [os.path.join(G, "pmc_%s%s_sq%d" % (run, tag, i)) for i in (1, 2)]
    if not all(os.path.isdir(d) for d in sq):  # tools/round_profiles.sh <run> bench: no counter passes in this run
        continue
    with open(os.path.join(P, "%s_pmc_sq_%s.txt" % (out, name)), "w") as fh:
        subprocess.run([sys.executable, summ] + sq, stdout=fh, check=True)
    tr = [os.path.join(G, "pmc_%s%s_%s" % (run, tag, c)) for c in ("fetch", "write")]
    with open(os.path.join(P, "%s_pmc_traffic%s.json" % (out, "" if name == "kem" else "_ntt")), "w") as fh:
        subprocess.run([sys.executable, summ, "--traffic-json"] + tr +
                       (["--note=bench.py --workload ntt --steps 3 --warmup 1 (2^20 polynomials per launch)"] if name == "ntt" else []), stdout=fh, check=True)
tr = [os.path.join(G, "pmc_%s_90s_%s" % (run, c)) for c in ("fetch", "write")]
if all(os.path.isdir(d) for d in tr):
    with open(os.path.join(P, "%s_pmc_traffic_90s.json" % out), "w") as fh:
        subprocess.run([sys.executable, summ, "--traffic-json", "--note=bench.py --workload kyber768-90s --steps 1 --warmup 1 --batch 524288 (2^19 instances per launch)"] + tr,
                       stdout=fh, check=True)
print("\n".join(sorted(os.listdir(P))))
