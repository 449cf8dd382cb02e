#!/usr/bin/env python3
"""Static opcode histogram of one kernel in a hipcc -S listing, split by the
basic-block labels so loop bodies can be told apart:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only kyber_hip.hip -o /tmp/kb/kyber.s
   python3 tools/isa_hist.py /tmp/kb/kyber.s <mangled-name-substring> [--blocks]"""
import collections
import sys

path, pat = sys.argv[1], sys.argv[2]
blocks = "--blocks" in sys.argv
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
tot = collections.Counter()
per = collections.OrderedDict()
cur = "entry"
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith("s_endpgm"):
        break
    if not t or t.startswith((";", ".")) and not t.startswith(".LBB"):
        continue
    if t.startswith(".LBB"):
        cur = t.split(":")[0]
        continue
    op = t.split()[0]
    tot[op] += 1
    per.setdefault(cur, collections.Counter())[op] += 1
print("total static instructions:", sum(tot.values()))
for k, v in tot.most_common(50):
    print("%6d %s" % (v, k))
if blocks:
    for b, c in per.items():
        n = sum(c.values())
        if n >= 40:
            print("\n%s: %d  " % (b, n) + ", ".join("%s %d" % kv for kv in c.most_common(12)))
