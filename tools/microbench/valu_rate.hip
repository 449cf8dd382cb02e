// valu_rate.hip -- per-instruction VALU issue cost on gfx950, measured with
// s_memtime (shader cycles) inside the kernel.  Each variant runs NCHAIN
// independent dependency chains of one instruction so latency is hidden.
// Output: cycles per wave-instruction per SIMD at 1, 2 and 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define KERNEL(NAME, ASM)                                                                          \
  __global__ void __launch_bounds__(1024) NAME(uint32_t* out, unsigned long long* cyc, int iters) { \
    uint32_t r0 = threadIdx.x, r1 = r0 * 3 + 1, r2 = r0 * 5 + 2, r3 = r0 * 7 + 3, r4 = r0 + 11,     \
             r5 = r0 ^ 0x55, r6 = r0 + 77, r7 = r0 * 9;                                             \
    uint32_t a = out[threadIdx.x & 63], b = out[(threadIdx.x + 1) & 63];                           \
    unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();                                     \
    unsigned long long t0 = __builtin_amdgcn_s_memtime();                                          \
    for (int i = 0; i < iters; i++) {                                                              \
      _Pragma("unroll") for (int u = 0; u < 8; u++) {                                              \
        asm volatile(ASM(0) ASM(1) ASM(2) ASM(3) ASM(4) ASM(5) ASM(6) ASM(7)                        \
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) \
                     : "v"(a), "v"(b));                                                            \
      }                                                                                            \
    }                                                                                              \
    unsigned long long t1 = __builtin_amdgcn_s_memtime();                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;             \
    unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();                                     \
    if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x + blockIdx.x] = rt1 - rt0; }  \
  }

#define A_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define A_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0x96\n"
#define A_BITOP3D2(n) "v_bitop3_b32 %" #n ", %" #n ", %8, %9 bitop3:0xd2\n"
#define A_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 13\n"
#define A_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define A_MULI24(n) "v_mul_i32_i24 %" #n ", %" #n ", %8\n"
#define A_MADI24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define A_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define A_MADI16(n) "v_mad_i32_i16 %" #n ", %" #n ", %8, %9\n"
#define A_PKADD(n) "v_pk_add_i16 %" #n ", %" #n ", %8\n"
#define A_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define A_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define A_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define A_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 3, %9\n"
#define A_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define A_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 4, 12\n"
#define A_ASHR(n) "v_ashrrev_i32 %" #n ", 16, %" #n "\n"
#define A_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define A_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define A_MULHI(n) "v_mul_hi_i32 %" #n ", %" #n ", %8\n"
#define A_XOR_E64(n) "v_xor_b32_e64 %" #n ", %" #n ", %8\n"
#define A_OR3(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n"
#define A_MOV(n) "v_mov_b32 %" #n ", %8\n"

KERNEL(k_xor, A_XOR)
KERNEL(k_xor_e64, A_XOR_E64)
KERNEL(k_bitop3, A_BITOP3)
KERNEL(k_bitop3d2, A_BITOP3D2)
KERNEL(k_alignbit, A_ALIGNBIT)
KERNEL(k_bfi, A_BFI)
KERNEL(k_or3, A_OR3)
KERNEL(k_muli24, A_MULI24)
KERNEL(k_madi24, A_MADI24)
KERNEL(k_mullo, A_MULLO)
KERNEL(k_mulhi, A_MULHI)
KERNEL(k_madi16, A_MADI16)
KERNEL(k_pkadd, A_PKADD)
KERNEL(k_pkmul, A_PKMUL)
KERNEL(k_add, A_ADD)
KERNEL(k_add3, A_ADD3)
KERNEL(k_lshladd, A_LSHLADD)
KERNEL(k_andor, A_ANDOR)
KERNEL(k_bfe, A_BFE)
KERNEL(k_ashr, A_ASHR)
KERNEL(k_perm, A_PERM)
KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_mov, A_MOV)


#define A_CMPCND(n) "v_cmp_gt_u32 vcc, %8, %" #n "\nv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n"
#define A_CMP(n) "v_cmp_gt_u32 vcc, %8, %" #n "\n"
#define A_CMPADDC(n) "v_cmp_gt_u32 vcc, %8, %" #n "\nv_addc_co_u32 %" #n ", vcc, 0, %" #n ", vcc\n"
#define A_CNDS(n) "v_cndmask_b32 %" #n ", %" #n ", %8, s[10:11]\n"
#define A_LSHL(n) "v_lshlrev_b32 %" #n ", 3, %" #n "\n"
#define A_LSHR(n) "v_lshrrev_b32 %" #n ", 3, %" #n "\n"
#define A_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define A_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define A_MIN(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define A_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define A_XORSDWA(n) "v_xor_b32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define A_ADDI16(n) "v_add_i16 %" #n ", %" #n ", %8\n"
#define A_MULU16(n) "v_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define A_MADU16(n) "v_mad_u16 %" #n ", %" #n ", %8, %9\n"
#define A_PKMAD(n) "v_pk_mad_i16 %" #n ", %" #n ", %8, %9\n"
#define A_PKASHR(n) "v_pk_ashrrev_i16 %" #n ", 3, %" #n "\n"
#define A_MULHIU(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"
#define A_SAD(n) "v_sad_u32 %" #n ", %" #n ", %8, %9\n"
#define A_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %8\n"
#define A_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %" #n ", %8\n"
#define A_DOT2(n) "v_dot2_i32_i16 %" #n ", %8, %9, %" #n "\n"
#define A_DOT4(n) "v_dot4_i32_i8 %" #n ", %8, %9, %" #n "\n"
#define A_ANDLIT(n) "v_and_b32 %" #n ", 0xabc, %" #n "\n"
#define A_ADDLIT(n) "v_add_u32 %" #n ", 0xfffff2ff, %" #n "\n"
#define A_ANDINL(n) "v_and_b32 %" #n ", 2, %" #n "\n"
KERNEL(k_andlit, A_ANDLIT)
KERNEL(k_addlit, A_ADDLIT)
KERNEL(k_andinl, A_ANDINL)
KERNEL(k_cmpcnd, A_CMPCND)
KERNEL(k_cmp, A_CMP)
KERNEL(k_cmpaddc, A_CMPADDC)
KERNEL(k_lshl, A_LSHL)
KERNEL(k_lshr, A_LSHR)
KERNEL(k_and, A_AND)
KERNEL(k_sub, A_SUB)
KERNEL(k_min, A_MIN)
KERNEL(k_med3, A_MED3)
KERNEL(k_xorsdwa, A_XORSDWA)
KERNEL(k_addi16, A_ADDI16)
KERNEL(k_mulu16, A_MULU16)
KERNEL(k_madu16, A_MADU16)
KERNEL(k_pkmad, A_PKMAD)
KERNEL(k_pkashr, A_PKASHR)
KERNEL(k_mulhiu, A_MULHIU)
KERNEL(k_sad, A_SAD)
KERNEL(k_bcnt, A_BCNT)
KERNEL(k_mbcnt, A_MBCNT)
KERNEL(k_dot2, A_DOT2)
KERNEL(k_dot4, A_DOT4)

#define A_FMUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define A_FADD(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define A_FFMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define A_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
#define A_RNDNE(n) "v_rndne_f32 %" #n ", %" #n "\n"
#define A_CVTFI(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define A_CVTIF(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define A_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n"
#define A_FRACT(n) "v_fract_f32 %" #n ", %" #n "\n"
KERNEL(k_fmul, A_FMUL)
KERNEL(k_fadd, A_FADD)
KERNEL(k_ffma, A_FFMA)
KERNEL(k_fmac, A_FMAC)
KERNEL(k_rndne, A_RNDNE)
KERNEL(k_cvtfi, A_CVTFI)
KERNEL(k_cvtif, A_CVTIF)
KERNEL(k_floor, A_FLOOR)
KERNEL(k_fract, A_FRACT)

#define A_FMAMIX(n) "v_fma_mix_f32 %" #n ", %8, %9, %" #n " op_sel_hi:[1,0,0]\n"
#define A_FMAMIXHI(n) "v_fma_mix_f32 %" #n ", %8, %9, %" #n " op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define A_CVTF16(n) "v_cvt_f32_f16 %" #n ", %" #n "\n"
#define A_CVTF16HI(n) "v_cvt_f32_f16_sdwa %" #n ", %" #n " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define A_CVTI16HI(n) "v_cvt_f32_i32_sdwa %" #n ", sext(%" #n ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define A_CVTPKRTZ(n) "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n"
#define A_CVTUB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define A_PKFMA(n) "v_pk_fma_f32 %[p" #n "], %[p" #n "], %[pa], %[pb]\n"
KERNEL(k_fmamix, A_FMAMIX)
KERNEL(k_fmamixhi, A_FMAMIXHI)
KERNEL(k_cvtf16, A_CVTF16)
KERNEL(k_cvtf16hi, A_CVTF16HI)
KERNEL(k_cvti16hi, A_CVTI16HI)
KERNEL(k_cvtpkrtz, A_CVTPKRTZ)
KERNEL(k_cvtub, A_CVTUB)

// 32 v_bitop3_b32 and 32 v_alignbit_b32 per 64 instructions, in runs of RUN: does changing between
// the full-rate and the half-rate class cost issue cycles?
template <int RUN>
__global__ void __launch_bounds__(1024) k_mix(uint32_t* out, unsigned long long* cyc, int iters) {
  uint32_t r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = threadIdx.x * (2 * j + 1) + j;
  uint32_t a = out[threadIdx.x & 63], b = out[(threadIdx.x + 1) & 63];
  unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 64; u++) {
      if ((u / RUN) & 1) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(r[u % 8]) : "v"(a));
      else asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[u % 8]) : "v"(a), "v"(b));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] ^ r[1] ^ r[2] ^ r[3] ^ r[4] ^ r[5] ^ r[6] ^ r[7];
  unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x + blockIdx.x] = rt1 - rt0; }
}

// 96 instructions per iteration on CH independent dependency chains (CH divides 96): how does the
// distance between an instruction and the one that consumes its result change the issue rate?
// OP 0: v_add_u32 (result -> src0), 1: v_add_u32 (result -> src1), 2: v_bitop3_b32, 3: v_alignbit_b32,
// 4: v_fma_f32
template <int CH, int OP>
__global__ void __launch_bounds__(1024) k_dep(uint32_t* out, unsigned long long* cyc, int iters) {
  uint32_t r[16];
#pragma unroll
  for (int j = 0; j < 16; j++) r[j] = threadIdx.x * (2 * j + 1) + j;
  uint32_t a = out[threadIdx.x & 63], b = out[(threadIdx.x + 1) & 63];
  unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters * 2 / 3; i++) {
#pragma unroll
    for (int u = 0; u < 96; u++) {
      if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[u % CH]) : "v"(a));
      if (OP == 1) asm volatile("v_add_u32 %0, %1, %0" : "+v"(r[u % CH]) : "v"(a));
      if (OP == 2) asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(r[u % CH]) : "v"(a), "v"(b));
      if (OP == 3) asm volatile("v_alignbit_b32 %0, %0, %1, 13" : "+v"(r[u % CH]) : "v"(a));
      if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[u % CH]) : "v"(a), "v"(b));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  uint32_t x = 0;
#pragma unroll
  for (int j = 0; j < 16; j++) x ^= r[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
  unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x + blockIdx.x] = rt1 - rt0; }
}
// the rejection parse's instruction pattern: per candidate and-literal, add-literal, shift, and-inline,
// cursor add (the cursor is one serial chain), 8 candidates per block on 8 value registers
template <bool LIT>
__global__ void __launch_bounds__(1024) k_parse(uint32_t* out, unsigned long long* cyc, int iters) {
  uint32_t r[8];
#pragma unroll
  for (int j = 0; j < 8; j++) r[j] = threadIdx.x * (2 * j + 1) + j;
  uint32_t cur = out[threadIdx.x & 63], m = 0xabc, q = 0xfffff2ffu;
  asm volatile("" : "+v"(m), "+v"(q));
  unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 12; u++) {   // 12 x 5 = 60 instructions + 4 fill = 64
      uint32_t d;
      if (LIT) {
        asm volatile("v_and_b32 %0, 0xabc, %1" : "=v"(d) : "v"(r[u % 8]));
        asm volatile("v_add_u32 %0, 0xfffff2ff, %0" : "+v"(d));
      } else {
        asm volatile("v_and_b32 %0, %2, %1" : "=v"(d) : "v"(r[u % 8]), "v"(m));
        asm volatile("v_add_u32 %0, %1, %0" : "+v"(d) : "v"(q));
      }
      asm volatile("v_lshrrev_b32 %0, 30, %0" : "+v"(d));
      asm volatile("v_and_b32 %0, 2, %0" : "+v"(d));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(cur) : "v"(d));
    }
#pragma unroll
    for (int u = 0; u < 4; u++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r[u]) : "v"(cur));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = r[0] ^ r[1] ^ r[2] ^ r[3] ^ cur;
  unsigned long long rt1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { cyc[blockIdx.x] = t1 - t0; cyc[gridDim.x + blockIdx.x] = rt1 - rt0; }
}

typedef void (*kfn)(uint32_t*, unsigned long long*, int);
struct V { const char* name; kfn f; };

int main() {
  V vs[] = {{"v_fma_mix_f32 lo", k_fmamix}, {"v_fma_mix_f32 hi", k_fmamixhi}, {"v_cvt_f32_f16", k_cvtf16}, {"v_cvt_f32_f16 W1", k_cvtf16hi},
            {"v_cvt_f32_i32 W1", k_cvti16hi}, {"v_cvt_pkrtz_f16", k_cvtpkrtz}, {"v_cvt_f32_ubyte1", k_cvtub},
            {"v_mul_f32", k_fmul}, {"v_add_f32", k_fadd}, {"v_fma_f32", k_ffma}, {"v_fmac_f32", k_fmac}, {"v_rndne_f32", k_rndne},
            {"v_cvt_f32_i32", k_cvtfi}, {"v_cvt_i32_f32", k_cvtif}, {"v_floor_f32", k_floor}, {"v_fract_f32", k_fract},
            {"v_xor_b32(e32)", k_xor}, {"v_xor_b32(e64)", k_xor_e64}, {"v_bitop3 0x96", k_bitop3}, {"v_bitop3 0xd2", k_bitop3d2},
            {"v_alignbit", k_alignbit}, {"v_bfi", k_bfi}, {"v_or3", k_or3}, {"v_mul_i32_i24", k_muli24}, {"v_mad_i32_i24", k_madi24},
            {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_i32", k_mulhi}, {"v_mad_i32_i16", k_madi16}, {"v_pk_add_i16", k_pkadd},
            {"v_pk_mul_lo_u16", k_pkmul}, {"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_lshl_add_u32", k_lshladd},
            {"v_and_or_b32", k_andor}, {"v_bfe_u32", k_bfe}, {"v_ashrrev_i32", k_ashr}, {"v_perm_b32", k_perm},
            {"v_cndmask", k_cndmask}, {"v_mov_b32", k_mov},
            {"cmp+cndmask(x2)", k_cmpcnd}, {"v_cmp_gt_u32", k_cmp}, {"cmp+addc(x2)", k_cmpaddc}, {"v_lshlrev_b32", k_lshl},
            {"v_lshrrev_b32", k_lshr}, {"v_and_b32", k_and}, {"v_sub_u32", k_sub}, {"v_min_u32", k_min}, {"v_med3_i32", k_med3},
            {"v_xor_b32_sdwa", k_xorsdwa}, {"v_add_i16", k_addi16}, {"v_mul_lo_u16", k_mulu16}, {"v_mad_u16", k_madu16},
            {"v_pk_mad_i16", k_pkmad}, {"v_pk_ashrrev_i16", k_pkashr}, {"v_mul_hi_u32", k_mulhiu}, {"v_sad_u32", k_sad},
            {"v_and_b32 literal", k_andlit}, {"v_add_u32 literal", k_addlit}, {"v_and_b32 inline", k_andinl},
            #define DEPS(NAME, OP) {NAME " dist 1", k_dep<1, OP>}, {NAME " dist 2", k_dep<2, OP>}, {NAME " dist 3", k_dep<3, OP>}, {NAME " dist 4", k_dep<4, OP>}, \
            {NAME " dist 6", k_dep<6, OP>}, {NAME " dist 8", k_dep<8, OP>}, {NAME " dist 12", k_dep<12, OP>}, {NAME " dist 16", k_dep<16, OP>},
            DEPS("add src0", 0) DEPS("add src1", 1) DEPS("bitop3", 2) DEPS("alignbit", 3) DEPS("fma", 4)
            {"parse mix literal", k_parse<true>}, {"parse mix vgpr", k_parse<false>},
            {"mix run 1", k_mix<1>}, {"mix run 2", k_mix<2>}, {"mix run 4", k_mix<4>}, {"mix run 8", k_mix<8>}, {"mix run 32", k_mix<32>},
            {"v_bcnt_u32_b32", k_bcnt}, {"v_mbcnt_lo", k_mbcnt}, {"v_dot2_i32_i16", k_dot2}, {"v_dot4_i32_i8", k_dot4}};
  uint32_t* out; unsigned long long* cyc;
  const int nblk = 512;
  CK(hipMalloc(&out, nblk * 1024 * 4 * 4)); CK(hipMemset(out, 1, nblk * 1024 * 4 * 4));
  CK(hipMalloc(&cyc, nblk * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 2000;  // x 64 instructions
  { int wk = 0; CK(hipDeviceGetAttribute(&wk, hipDeviceAttributeWallClockRate, 0)); printf("wall clock rate %d kHz\n", wk); }
  printf("per config: memtime ticks per wave-instr per SIMD | shader MHz from memtime/memrealtime | T lane-op/s by wall clock\n");
  for (auto& v : vs) {
    printf("%-18s", v.name);
    for (int cfg = 1; cfg < 3; cfg++) {
      int threads = cfg == 0 ? 256 : cfg == 1 ? 512 : 1024;
      int blocks = cfg == 3 ? 512 : 256;
      hipLaunchKernelGGL(v.f, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(v.f, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      std::vector<unsigned long long> h(2 * blocks);
      CK(hipMemcpy(h.data(), cyc, 2 * blocks * 8, hipMemcpyDeviceToHost));
      double s = 0, rt = 0; for (int i = 0; i < blocks; i++) { s += (double)h[i]; rt += (double)h[blocks + i]; }
      s /= blocks; rt /= blocks;
      int wps = threads / 256 * (blocks / 256);
      double laneops = (double)blocks * threads * iters * 64.0;
      printf(" | %dw %6.3f %5.0fMHz %6.2fT", wps, s / ((double)iters * 64 * wps), s / rt * 100.0, laneops / (ms * 1e-3) * 1e-12);
    }
    printf("\n");
  }
  return 0;
}
