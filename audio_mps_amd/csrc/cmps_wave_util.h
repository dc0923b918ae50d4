// Device helpers shared by the wave-per-clip kernels (cmps_wave.hip, cmps_wave2.hip): lane layout, packed complex
// multiply-accumulate chains, cross-half combines, DPP wave reduction, inline-asm LDS traffic with counted waits,
// chunk staging.  See the header comment of cmps_wave.hip for the layout these operate on.
#pragma once
#include <type_traits>

#include "cmps_internal.h"

namespace cmps {

namespace {

constexpr int DPW = 32;    // padded bond dimension of this variant
constexpr int WAVES = 4;   // waves (clips) per workgroup: one per SIMD of a CU
constexpr int CH = 64;     // steps per chunk of per-step scalars (one step per lane); forward table staging
constexpr int PE_LD = 65;  // row stride (floats) of the forward's per-lane product buffer: conflict-free both ways
constexpr int CHB = 32;    // steps per staged chunk in the reverse sweep (three tables share the LDS)

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v2f mk2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f ld2(const float2* p) { const float2 t = *p; return mk2(t.x, t.y); }
__device__ __forceinline__ v2f lo2(v4f q) { return __builtin_shufflevector(q, q, 0, 1); }
__device__ __forceinline__ v2f hi2(v4f q) { return __builtin_shufflevector(q, q, 2, 3); }

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// ---- packed complex multiply-accumulate: acc += a * b  =  [a * Re b]  +  [i a * Im b] ----
__device__ __forceinline__ void pkmul_bl(v2f& acc, v2f a, v2f b) {   // acc = (a.x, a.y) * (b.x, b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pkfma_bl(v2f& acc, v2f a, v2f b) {   // acc += (a.x, a.y) * (b.x, b.x)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pkfma_bh(v2f& acc, v2f a, v2f b) {   // acc += (-a.y, a.x) * (b.y, b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void pkfma_bh_conj(v2f& acc, v2f a, v2f b) {  // acc += (a.y, -a.x) * (b.y, b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "+v"(acc) : "v"(a), "v"(b));
}
// (one asm statement each: hipcc pads every statement boundary)
__device__ __forceinline__ v2f cmul2(v2f a, v2f b) {            // a * b
    v2f acc;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=&v"(acc) : "v"(a), "v"(b));
    return acc;
}
__device__ __forceinline__ v2f cmul2_conj_b(v2f a, v2f b) {     // a * conj(b) = a*Re b - i a*Im b
    v2f acc;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=&v"(acc) : "v"(a), "v"(b));
    return acc;
}

// ---- mat-vec cores.  One asm statement per 8 complex multiply-accumulates (16 packed instructions):
// hipcc pads every asm statement boundary with an s_nop, which costs a full issue slot for a lone wave, so
// the chains are emitted as a few large blocks.  Dependent accumulation inside a block is free (see header).
// operand numbering: CM(acc, m, b): acc += M_m * b   (complex, two packed FMAs)
#define CM_FIRST(acc, m, b)                                                                   \
    "v_pk_mul_f32 %" #acc ", %" #m ", %" #b " op_sel:[0,0] op_sel_hi:[1,0]\n\t"                 \
    "v_pk_fma_f32 %" #acc ", %" #m ", %" #b ", %" #acc " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"
#define CM(acc, m, b)                                                                         \
    "v_pk_fma_f32 %" #acc ", %" #m ", %" #b ", %" #acc " op_sel:[0,0,0] op_sel_hi:[1,0,1]\n\t"   \
    "v_pk_fma_f32 %" #acc ", %" #m ", %" #b ", %" #acc " op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\t"

// this half's 16 columns of (M v), one chain
__device__ __forceinline__ v2f mv1(const v2f (&M)[16], const v4f (&q)[8]) {
    v2f acc;
    asm(CM_FIRST(0, 1, 9) CM(0, 2, 10) CM(0, 3, 11) CM(0, 4, 12) CM(0, 5, 13) CM(0, 6, 14) CM(0, 7, 15) CM(0, 8, 16)
        : "=&v"(acc)
        : "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(M[4]), "v"(M[5]), "v"(M[6]), "v"(M[7]),
          "v"(lo2(q[0])), "v"(hi2(q[0])), "v"(lo2(q[1])), "v"(hi2(q[1])), "v"(lo2(q[2])), "v"(hi2(q[2])),
          "v"(lo2(q[3])), "v"(hi2(q[3])));
    asm(CM(0, 1, 9) CM(0, 2, 10) CM(0, 3, 11) CM(0, 4, 12) CM(0, 5, 13) CM(0, 6, 14) CM(0, 7, 15) CM(0, 8, 16)
        : "+v"(acc)
        : "v"(M[8]), "v"(M[9]), "v"(M[10]), "v"(M[11]), "v"(M[12]), "v"(M[13]), "v"(M[14]), "v"(M[15]),
          "v"(lo2(q[4])), "v"(hi2(q[4])), "v"(lo2(q[5])), "v"(hi2(q[5])), "v"(lo2(q[6])), "v"(hi2(q[6])),
          "v"(lo2(q[7])), "v"(hi2(q[7])));
    return acc;
}
// the same chain in two halves, so that the first can start as soon as the first four broadcast reads have landed
__device__ __forceinline__ void mv1_lo(const v2f (&M)[16], const v4f (&q)[8], v2f& acc) {
    asm(CM_FIRST(0, 1, 9) CM(0, 2, 10) CM(0, 3, 11) CM(0, 4, 12) CM(0, 5, 13) CM(0, 6, 14) CM(0, 7, 15) CM(0, 8, 16)
        : "=&v"(acc)
        : "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(M[4]), "v"(M[5]), "v"(M[6]), "v"(M[7]),
          "v"(lo2(q[0])), "v"(hi2(q[0])), "v"(lo2(q[1])), "v"(hi2(q[1])), "v"(lo2(q[2])), "v"(hi2(q[2])),
          "v"(lo2(q[3])), "v"(hi2(q[3])));
}
__device__ __forceinline__ void mv1_hi(const v2f (&M)[16], const v4f (&q)[8], v2f& acc) {
    asm(CM(0, 1, 9) CM(0, 2, 10) CM(0, 3, 11) CM(0, 4, 12) CM(0, 5, 13) CM(0, 6, 14) CM(0, 7, 15) CM(0, 8, 16)
        : "+v"(acc)
        : "v"(M[8]), "v"(M[9]), "v"(M[10]), "v"(M[11]), "v"(M[12]), "v"(M[13]), "v"(M[14]), "v"(M[15]),
          "v"(lo2(q[4])), "v"(hi2(q[4])), "v"(lo2(q[5])), "v"(hi2(q[5])), "v"(lo2(q[6])), "v"(hi2(q[6])),
          "v"(lo2(q[7])), "v"(hi2(q[7])));
}
// a quarter of the chain (4 entries, 8 packed FMAs): lets a caller put something between two quarters
template <bool FIRST>
__device__ __forceinline__ void mv1_quarter(const v2f& m0, const v2f& m1, const v2f& m2, const v2f& m3, v4f qa, v4f qb, v2f& acc) {
    if constexpr (FIRST)
        asm(CM_FIRST(0, 1, 5) CM(0, 2, 6) CM(0, 3, 7) CM(0, 4, 8)
            : "=&v"(acc) : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(lo2(qa)), "v"(hi2(qa)), "v"(lo2(qb)), "v"(hi2(qb)));
    else
        asm(CM(0, 1, 5) CM(0, 2, 6) CM(0, 3, 7) CM(0, 4, 8)
            : "+v"(acc) : "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(lo2(qa)), "v"(hi2(qa)), "v"(lo2(qb)), "v"(hi2(qb)));
}
// two matrices applied to the same vector, chains interleaved; in two halves so that the first can start as
// soon as the first four broadcast reads have landed
__device__ __forceinline__ void mv2_lo(const v2f (&MA)[16], const v2f (&MB)[16], const v4f (&q)[8], v2f& accA, v2f& accB) {
    asm(CM_FIRST(0, 2, 18) CM_FIRST(1, 10, 18) CM(0, 3, 19) CM(1, 11, 19) CM(0, 4, 20) CM(1, 12, 20) CM(0, 5, 21) CM(1, 13, 21)
        CM(0, 6, 22) CM(1, 14, 22) CM(0, 7, 23) CM(1, 15, 23) CM(0, 8, 24) CM(1, 16, 24) CM(0, 9, 25) CM(1, 17, 25)
        : "=&v"(accA), "=&v"(accB)
        : "v"(MA[0]), "v"(MA[1]), "v"(MA[2]), "v"(MA[3]), "v"(MA[4]), "v"(MA[5]), "v"(MA[6]), "v"(MA[7]),
          "v"(MB[0]), "v"(MB[1]), "v"(MB[2]), "v"(MB[3]), "v"(MB[4]), "v"(MB[5]), "v"(MB[6]), "v"(MB[7]),
          "v"(lo2(q[0])), "v"(hi2(q[0])), "v"(lo2(q[1])), "v"(hi2(q[1])), "v"(lo2(q[2])), "v"(hi2(q[2])),
          "v"(lo2(q[3])), "v"(hi2(q[3])));
}
__device__ __forceinline__ void mv2_hi(const v2f (&MA)[16], const v2f (&MB)[16], const v4f (&q)[8], v2f& accA, v2f& accB) {
    asm(CM(0, 2, 18) CM(1, 10, 18) CM(0, 3, 19) CM(1, 11, 19) CM(0, 4, 20) CM(1, 12, 20) CM(0, 5, 21) CM(1, 13, 21)
        CM(0, 6, 22) CM(1, 14, 22) CM(0, 7, 23) CM(1, 15, 23) CM(0, 8, 24) CM(1, 16, 24) CM(0, 9, 25) CM(1, 17, 25)
        : "+v"(accA), "+v"(accB)
        : "v"(MA[8]), "v"(MA[9]), "v"(MA[10]), "v"(MA[11]), "v"(MA[12]), "v"(MA[13]), "v"(MA[14]), "v"(MA[15]),
          "v"(MB[8]), "v"(MB[9]), "v"(MB[10]), "v"(MB[11]), "v"(MB[12]), "v"(MB[13]), "v"(MB[14]), "v"(MB[15]),
          "v"(lo2(q[4])), "v"(hi2(q[4])), "v"(lo2(q[5])), "v"(hi2(q[5])), "v"(lo2(q[6])), "v"(hi2(q[6])),
          "v"(lo2(q[7])), "v"(hi2(q[7])));
}

__device__ __forceinline__ float rdlane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---- cross-half: (partial.x, partial.y) of both halves -> split-layout total ----
__device__ __forceinline__ float swapadd(float px, float py) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(px), __float_as_uint(py), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);     // half 0: sum of x's; half 1: sum of y's
}
// The same directly behind an asm block that wrote px / py (the mat-vec chains): hipcc's hazard recogniser does not see VALU writes inside
// inline asm and left ONE instruction between the chain's last v_pk_fma_f32 and the exchange in k_bwd_wave / k_fwd_wave16, where its own
// code keeps two wait states (s_nop 1) -- found by scripts/check_mfma_hazards.py's round-5 check.  No wrong result was ever observed (a
// lone wave issues every ~5 cycles), but the distance is now written out.
__device__ __forceinline__ float swapadd_after_asm(float px, float py) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(px), "+v"(py));
    return px + py;
}
// split value x -> osig (see header): half 0 gets the partner's value, half 1 minus the partner's value
__device__ __forceinline__ float osig_of(float x, bool hbit) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return hbit ? -__uint_as_float(r[0]) : __uint_as_float(r[1]);
}

// ---- wave reduction ----
template <int CTRL>
__device__ __forceinline__ float dpp_add_row(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float sum64(float x) {   // sum over all 64 lanes, uniform (SGPR) result
    x = dpp_add_row<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_add_row<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_add_row<0x141>(x);   // row_half_mirror
    x = dpp_add_row<0x140>(x);   // row_mirror        -> every lane holds its 16-lane row's sum
    // row_bcast15 into rows 1,3 then row_bcast31 into rows 2,3 (the s_nop covers the VALU-write -> DPP-read
    // hazard, which hipcc does not see inside an asm statement)
    asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
    return rdlane(x, 63);
}

__device__ __forceinline__ float rsq_nr(float m) {   // 1/sqrt(m): v_rsq_f32 + one Newton step
    const float r = __builtin_amdgcn_rsqf(m);
    return r * (1.5f - 0.5f * m * r * r);
}

// ---- LDS traffic of the inner loops, hidden from hipcc's waitcnt bookkeeping on purpose ----
// broadcast: every lane writes its 4-byte split value; this half then reads its 16 complex entries and one
// 8-byte table entry (rho).  Outputs are valid only after a matching lds_wait*.
__device__ __forceinline__ void bcast_issue(unsigned wr, unsigned rd, float mine, v4f (&o)[8]) {
    asm volatile("ds_write_b32 %8, %9\n\t"
                 "ds_read_b128 %0, %10\n\tds_read_b128 %1, %10 offset:16\n\t"
                 "ds_read_b128 %2, %10 offset:32\n\tds_read_b128 %3, %10 offset:48\n\t"
                 "ds_read_b128 %4, %10 offset:64\n\tds_read_b128 %5, %10 offset:80\n\t"
                 "ds_read_b128 %6, %10 offset:96\n\tds_read_b128 %7, %10 offset:112"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(wr), "v"(mine), "v"(rd) : "memory");
}
__device__ __forceinline__ void bcast_issue_tab(unsigned wr, unsigned rd, float mine, unsigned tab,
                                                v4f (&o)[8], v2f& t) {
    asm volatile("ds_write_b32 %9, %10\n\t"
                 "ds_read_b128 %0, %11\n\tds_read_b128 %1, %11 offset:16\n\t"
                 "ds_read_b128 %2, %11 offset:32\n\tds_read_b128 %3, %11 offset:48\n\t"
                 "ds_read_b128 %4, %11 offset:64\n\tds_read_b128 %5, %11 offset:80\n\t"
                 "ds_read_b128 %6, %11 offset:96\n\tds_read_b128 %7, %11 offset:112\n\t"
                 "ds_read_b64 %8, %12"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(t)
                 : "v"(wr), "v"(mine), "v"(rd), "v"(tab) : "memory");
}
// The same with the wait inside the statement: for a broadcast whose registers cross a loop back-edge before the counted wait of the next
// step.  hipcc may copy loop-carried registers at the back-edge, and it takes an asm statement's outputs for valid when the statement
// ends: in k_fwd_wave2<no stash> it copied the broadcast registers between the chunk-end issue and the next chunk's first wait
// (scripts/check_mfma_hazards.py, round 5; the 50 instructions in between had always covered the LDS latency).
__device__ __forceinline__ void bcast_issue_tab_sync(unsigned wr, unsigned rd, float mine, unsigned tab,
                                                     v4f (&o)[8], v2f& t) {
    asm volatile("ds_write_b32 %9, %10\n\t"
                 "ds_read_b128 %0, %11\n\tds_read_b128 %1, %11 offset:16\n\t"
                 "ds_read_b128 %2, %11 offset:32\n\tds_read_b128 %3, %11 offset:48\n\t"
                 "ds_read_b128 %4, %11 offset:64\n\tds_read_b128 %5, %11 offset:80\n\t"
                 "ds_read_b128 %6, %11 offset:96\n\tds_read_b128 %7, %11 offset:112\n\t"
                 "ds_read_b64 %8, %12\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(t)
                 : "v"(wr), "v"(mine), "v"(rd), "v"(tab) : "memory");
}
// the same with a compile-time offset on the table address
template <int TOFF>
__device__ __forceinline__ void bcast_issue_tab_off(unsigned wr, unsigned rd, float mine, unsigned tab, v4f (&o)[8], v2f& t) {
    asm volatile("ds_write_b32 %9, %10\n\t"
                 "ds_read_b128 %0, %11\n\tds_read_b128 %1, %11 offset:16\n\t"
                 "ds_read_b128 %2, %11 offset:32\n\tds_read_b128 %3, %11 offset:48\n\t"
                 "ds_read_b128 %4, %11 offset:64\n\tds_read_b128 %5, %11 offset:80\n\t"
                 "ds_read_b128 %6, %11 offset:96\n\tds_read_b128 %7, %11 offset:112\n\t"
                 "ds_read_b64 %8, %12 offset:%13"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(t)
                 : "v"(wr), "v"(mine), "v"(rd), "v"(tab), "n"(TOFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_write32_imm(unsigned wr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(wr), "v"(v), "n"(OFF) : "memory");
}
// row read (no write): this half's 16 entries of a vector written earlier
__device__ __forceinline__ void rows_issue(unsigned rd, v4f (&o)[8]) {
    asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:16\n\t"
                 "ds_read_b128 %2, %8 offset:32\n\tds_read_b128 %3, %8 offset:48\n\t"
                 "ds_read_b128 %4, %8 offset:64\n\tds_read_b128 %5, %8 offset:80\n\t"
                 "ds_read_b128 %6, %8 offset:96\n\tds_read_b128 %7, %8 offset:112"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(rd) : "memory");
}
__device__ __forceinline__ void lds_write32(unsigned wr, float v) {
    asm volatile("ds_write_b32 %0, %1" : : "v"(wr), "v"(v) : "memory");
}
// what the reverse step's off-chain stage reads: the lane's own (y, H y) of the staged row (8 B), rho (8 B), and
// the step's scalar row (2 x 16 B, the same address for every lane)
__device__ __forceinline__ void own_issue(unsigned ay, unsigned ar, unsigned as, v2f& yh, v2f& t, v4f& c0, v4f& c1) {
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %6 offset:16"
                 : "=&v"(yh), "=&v"(t), "=&v"(c0), "=&v"(c1) : "v"(ay), "v"(ar), "v"(as) : "memory");
}
// the same with compile-time row offsets on per-octet base addresses (keeps the address arithmetic out of the step)
template <int OY, int OR, int OS>
__device__ __forceinline__ void own_issue_off(unsigned ay, unsigned ar, unsigned as, v2f& yh, v2f& t, v4f& c0, v4f& c1) {
    asm volatile("ds_read_b64 %0, %4 offset:%7\n\tds_read_b64 %1, %5 offset:%8\n\tds_read_b128 %2, %6 offset:%9\n\t"
                 "ds_read_b128 %3, %6 offset:%10"
                 : "=&v"(yh), "=&v"(t), "=&v"(c0), "=&v"(c1) : "v"(ay), "v"(ar), "v"(as), "n"(OY), "n"(OR), "n"(OS), "n"(OS + 16)
                 : "memory");
}
// LDS operations of one wave complete in order, so "at most N outstanding" retires everything issued
// before the last N; extra operations hipcc may have in flight only make the wait stricter.
template <int N>
__device__ __forceinline__ void lds_wait(v4f (&o)[8]) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7])
                 : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_lo(v4f (&o)[8]) {     // first four reads of a broadcast
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_hi(v4f (&o)[8]) {     // last four reads
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_hi_t(v4f (&o)[8], v2f& t) {   // last four reads + table entry
    asm volatile("s_waitcnt lgkmcnt(%5)" : "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(t) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_all(v4f (&a)[8], v4f (&b)[8], v2f& t) {
    asm volatile("s_waitcnt lgkmcnt(%17)"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(t)
                 : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_own(v2f& yh, v2f& t, v4f& c0, v4f& c1) {
    asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(yh), "+v"(t), "+v"(c0), "+v"(c1) : "n"(N) : "memory");
}

// ---- chunk staging: 4*NQ table rows of 256 B (16 float4 each) global -> registers -> LDS ----
template <int NQ>
__device__ __forceinline__ void stage_load(const float4* __restrict__ tab, int row0, int max_row, int lane,
                                           v4f (&r)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = q * 64 + lane;
        int row = row0 + (e >> 4);
        row = row < max_row ? row : max_row;
        const float4 t = tab[(size_t)row * 16 + (e & 15)];
        r[q] = v4f{t.x, t.y, t.z, t.w};
    }
}
// same for 512-B rows (32 float4 each): 2*NQ rows
// (stride: rows between consecutive steps -- 1, or the rank when the rows of a RhoCMPS column are read from the [clip][step][column] stash)
template <int NQ>
__device__ __forceinline__ void stage_load512(const float4* __restrict__ tab, int row0, int max_row, int lane,
                                              v4f (&r)[NQ], int stride = 1) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = q * 64 + lane;
        int row = row0 + (e >> 5);
        row = row < max_row ? row : max_row;
        const float4 t = tab[(size_t)row * stride * 32 + (e & 31)];
        r[q] = v4f{t.x, t.y, t.z, t.w};
    }
}
template <int NQ>
__device__ __forceinline__ void stage_commit(float4* lds, int lane, const v4f (&r)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) lds[q * 64 + lane] = make_float4(r[q].x, r[q].y, r[q].z, r[q].w);
}

// The four waves of a workgroup run the same instruction stream at the same pace; started together they hit
// the LDS with their 8-KB broadcast bursts at the same moment, every step.  A one-off start offset of a
// quarter step per wave keeps the bursts apart for the whole scan (measured: -2.5 % forward time).
__device__ __forceinline__ void stagger(int w) {
    const int wu = __builtin_amdgcn_readfirstlane(w);
    if (wu & 1) __builtin_amdgcn_s_sleep(4);
    if (wu & 2) { __builtin_amdgcn_s_sleep(4); __builtin_amdgcn_s_sleep(4); }
}

// per-chunk scalar stash: [B][NC][2][64] floats: n_k (true |y_k|^2), e_k, one step per lane
__device__ __forceinline__ size_t scal_off(int b, int NC, int c) { return ((size_t)b * NC + c) * 128; }

}  // namespace

}  // namespace cmps
