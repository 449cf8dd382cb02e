"""CPU: every product kernel of libkyber_hip.so fits its registers -- no VGPR/SGPR spills, no
scratch (private segment).  Read from the gfx950 assembly listings of the two HIP translation units
(build/*.gfx950.s, written by __graft_entry__.build(); compiled here if stale).  A spill in a hot
kernel is silent otherwise: round 2's k_enc_arith<Par<3>, CMP> wrote 344 MB of scratch per launch."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

FIELDS = ("vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")


def test_no_kernel_spills_or_uses_scratch():
    import __graft_entry__ as g
    seen = 0
    for path in g.device_asm():
        # kernel descriptors come as YAML blocks whose fields are sorted by name: read them per kernel
        text = open(path).read()
        blocks = re.split(r"\n  - \.agpr_count:", text)[1:]
        assert blocks, "no kernel metadata in %s" % path
        for b in blocks:
            name = re.search(r"\.name:\s+(\S+)", b).group(1)
            if "2ky" not in name:   # namespace ky: the product's kernels
                continue
            seen += 1
            for f in FIELDS:
                v = int(re.search(r"\.%s:\s+(\d+)" % f, b).group(1))
                assert v == 0, "%s: %s = %d" % (name, f, v)
    assert seen > 100, "only %d kernels found" % seen
