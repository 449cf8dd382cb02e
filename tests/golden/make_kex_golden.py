#!/usr/bin/env python3
"""Writes tests/golden/kex.json: sha256 over every output of the KEX scenario
(tests/kex_scenario.py, = ref/test_kex.c with explicit coins) computed by THE
REFERENCE ITSELF (oracle/_ref: the reference's ref/ sources compiled in place),
n = 16 instances, seed 4242 + 10 k + nineties, for all six parameter sets.

    python3 tests/golden/make_kex_golden.py      # needs /root/reference (build container only)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import kex_scenario  # noqa: E402
import oracle_lib  # noqa: E402

N = 16
out = {"n": N, "seed": "4242 + 10*k + nineties", "generator": "oracle/_ref libkyber_ref_* (reference ref/ sources)", "sha256": {}}
for k in (2, 3, 4):
    for n90 in (0, 1):
        ref = oracle_lib.RefLib("ref", k, n90)
        res = kex_scenario.run(ref.keypair, ref.kex(), kex_scenario.coins_for(N, 4242 + 10 * k + n90))
        kex_scenario.check_agreement(res)
        out["sha256"]["Kyber%d%s" % (256 * k, "-90s" if n90 else "")] = kex_scenario.digest(res)
with open(os.path.join(HERE, "kex.json"), "w") as fh:
    json.dump(out, fh, indent=1)
    fh.write("\n")
print(json.dumps(out["sha256"], indent=1))
