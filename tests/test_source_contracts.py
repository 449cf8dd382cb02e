"""CPU: contracts of the kernel sources that only a comment enforced.

keccak.h's phase lock puts raw `s_barrier`s inside the permutation (about 120 per Keccak-f).  That is
sound only while no kernel that permutes ALSO uses a workgroup barrier for data: a wavefront inside a
permutation would answer another wavefront's data barrier -- silent corruption, not a hang.  So: a
source file that calls the permutation must not contain __syncthreads(), and files that do use
__syncthreads() must not call the permutation."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "kyber_amd", "csrc")
PERMUTES = re.compile(r"\bkeccak_f1600\s*(<[^>]*>)?\s*\(|\bkeccak_absorb_words\s*<|\bsha3_256_32\(|\bsha3_512_64\(|\bshake256_64\(")
BARRIER = re.compile(r"__syncthreads\s*\(|__builtin_amdgcn_s_barrier")


def _code(path):
    """source text without comments"""
    text = open(path).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def test_no_kernel_source_mixes_the_phase_lock_with_a_data_barrier():
    users = []
    for path in sorted(glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.inc"))):
        code = _code(path)
        name = os.path.basename(path)
        if name == "keccak.h":
            assert not BARRIER.search(code), "keccak.h itself must not use a data barrier"
            continue
        if PERMUTES.search(code):
            users.append(name)
            assert not BARRIER.search(code), "%s calls the phase-locked permutation and uses a workgroup barrier" % name
    assert "kem_kernels.h" in users
    # the files that do use data barriers (LDS tables, transposes) never permute, and the translation
    # unit of the poly-layout kernels does not even include keccak.h
    arith = _code(os.path.join(CSRC, "kyber_arith.hip")) + _code(os.path.join(CSRC, "arith_kernels.h")) + _code(os.path.join(CSRC, "wpoly.h"))
    assert "keccak.h" not in arith and "kem_kernels" not in arith


def test_no_kernel_permutes_and_builds_a_shared_table():
    """per kernel: the AES kernels set their LDS tables up behind a workgroup barrier (aes.h
    aes_table_init); none of them may call the permutation, and no permuting kernel may call a
    helper that synchronises the workgroup"""
    syncing_helpers = re.compile(r"\baes_table_init\s*\(|\bf_twiddles_to_lds\s*\(|\btwiddles_to_lds\s*\(|__syncthreads\s*\(")
    seen = 0
    for name in ("kem_kernels.h", "kem_kernels_90s.h"):
        pieces = _code(os.path.join(CSRC, name)).split("__global__")[1:]
        for body in pieces:
            seen += 1
            kernel = re.search(r"\b(k\w+)\s*\(", body).group(1)
            assert not (PERMUTES.search(body) and syncing_helpers.search(body)), kernel
    assert seen >= 20
