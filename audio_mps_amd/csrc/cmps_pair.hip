// 32 < D <= 128 ("bf16 with fp32 accumulate", BASELINE configs[4] is D = 128): MFMA kernels, one workgroup per PAIR of
// clips, instantiated for the padded bond dimensions 64, 96 and 128 (components >= D are zero padding).
//
// At D = 128 and B = 512 there are two clips per CU and a matrix is 128 KB in complex64: nothing like the wave-per-clip
// layout fits, and a GEMM tile over clips would be two columns wide.  A complex mat-vec y = M u is the real product
// [M_re | M_im] (D x 2D)  times a 2D-vector, once with the form  c0 = [u_re; -u_im]  (-> Re y) and once with
// c1 = [u_im; u_re]  (-> Im y): with TWO clips there are FOUR such forms.  Round 4 runs it on v_mfma_f32_16x16x32_bf16 with
// the roles turned round (rounds 1-3 used the batched 4x4x4 MFMA: twice the instructions, a K split over the lane halves
// that had to be summed with permlane swaps, and a lone wave issues it every 12 cycles instead of its 8):
//
//   D (16 x 16) = A (16 x 32) B (32 x 16):   A rows = vector forms, B columns = sixteen ROWS of the matrix, K = 32 entries of
//   the real form.  A rows {0, 4, 8, 12} carry the four forms: row 4 f lands in accumulator register 0 of the lanes
//   16 f .. 16 f + 15, so after the K loop register 0 of lane l IS  (M u)[row (l & 15)], form (l >> 4)  -- K summed by the
//   matrix pipe, no cross-lane reduction, no compaction (scripts/ubench/mfma16_matvec.hip checks the layout with integer
//   data and times the instruction: 16 cycles back to back from a lone wave).  A rows {1, 5, 9, 13} carry the Re <-> Im
//   PARTNER of each form and land in register 1 of the same lanes: the complex rotations of the chain (rho y, conj(rho) g)
//   find both components of their row in-lane; the other eight A rows are unused.
//
//   workgroup = D / 32 chain waves (+ as many loss waves in the forward); wave w owns rows 32 w .. 32 w + 31 as TWO tiles
//   (tile 0 = the even rows, tile 1 = the odd rows: a lane's two values are adjacent rows, one packed 4-byte LDS store).
//   lane l = (j = l & 15, f = l >> 4): rows ia = 32 w + 2 j and ia + 1, component (f & 1 ? Im : Re) of clip f >> 1 -- every
//     (row, component, clip) lives in exactly one lane; its Re <-> Im partner sits 16 lanes away (v_permlane16_swap).
//   B operand (resident: AGPRs in the reverse scan; the forward has no AGPR operand at all, so hipcc selects the VGPR form of
//     the MFMA and its lone chain wave uses all 256 registers as one file): 8 bf16 = columns 32 t + 8 (l >> 4) .. + 7 of  [M_re | M_im]  in row 32 w + 2 (l & 15) + tile,
//     D / 16 K-steps t, two tiles: D / 2 registers per matrix and lane.
//   A operand: lane l reads 16 bytes of form (l >> 2) & 3, or of its partner in lanes with (l & 3) == 1 (the other two lanes
//     of a quad feed unused A rows and read the form's address) at K = 32 t + 8 (l >> 4): D / 16 ds_read_b128 per broadcast vector and lane, shared by both matrices.
//   Per step: 4 D / 16 MFMAs on the chain (R ut, Q ut; 512 matrix-pipe cycles at D = 128), ONE workgroup barrier.
//   The identity part of y = ut + Q ut + s R ut stays float32; only the small correction goes through bf16.
// Arithmetic restated (with the same rounding points) by oracle/cmps_oracle.py::psi_bf16_scan.
#include "cmps_internal.h"
#include "cmps_grad_gemm.h"

namespace cmps {

namespace {

constexpr int PCH = 64;      // steps per chunk of per-step scalars
// Everything below is templated on the padded bond dimension D (64, 96 or 128): D / 32 waves own 32 rows each, a mat-vec pair is
// 4 D / 16 MFMA instructions per wave, a broadcast vector is D / 16 16-byte reads per lane.

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rne(float f) {
    unsigned u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
// (lo, hi) -> packed bf16x2, round to nearest even
__device__ __forceinline__ unsigned pk_bf16(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float rdl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt, i.e. waits every step for the
// stash row's store to reach L2 (~1.5 us; measured: it was 80 % of the step)
__device__ __forceinline__ void lds_barrier() {
#if defined(CMPS_DIAG) && defined(PABL_NO_BARRIER)    // diagnostic builds only (scripts/ablate.py): results are wrong
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}
// sum over the two lane halves, both operands (see cmps_wave_util.h::swapadd); the loss waves' row halves
__device__ __forceinline__ float half_add(float a0, float a1) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a0), __float_as_uint(a1), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
template <int CTRL>
__device__ __forceinline__ unsigned dpp_movu(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// the value the lane 16 away holds (the Re <-> Im partner of the same row and clip): v_permlane16_swap of x with itself leaves
// the even rows' values in r[0] and the odd rows' values in r[1], in both rows of a pair
__device__ __forceinline__ float partner16(float x, bool odd) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(odd ? r[0] : r[1]);
}
// sum over the 16 lanes of a row (one clip, one component); every lane receives the total
__device__ __forceinline__ float row_sum16(float x) {
    x += dpp_mov<0x128>(x);       // row_ror:8
    x += dpp_mov<0x124>(x);       // row_ror:4
    x += dpp_mov<0x122>(x);       // row_ror:2
    x += dpp_mov<0x121>(x);       // row_ror:1
    return x;
}
template <int W>
__device__ __forceinline__ float sum4(const float __attribute__((ext_vector_type(4))) & t) {   // the first W entries
    if constexpr (W == 4) return (t.x + t.y) + (t.z + t.w);
    else if constexpr (W == 3) return (t.x + t.y) + t.z;
    else return t.x + t.y;
}
// sum over the two clips (lanes l and l ^ 32), in every lane
__device__ __forceinline__ float both_clips(float x) { return half_add(x, x); }

// LDS image of one broadcast vector for both clips: arrays re | im | -im | dummy, each [2 clips][D] bf16.
// Rows are padded by 32 B: the sixteen distinct 16-byte pieces a ds_read_b128 of the A operand touches -- four forms (four
// different rows at the same offset) x four K groups (16 bytes apart) -- then fall into sixteen different 4-bank groups
// (row r starts 8 r banks in; with the 16-byte padding of rounds 1-3 form f + 1 at K group g met form f at K group g + 1).
template <int D>
struct PairLds {
    static constexpr int VROW = D * 2 + 32;                           // bytes per (array, clip) row
    static constexpr int VEC_BYTES = 8 * VROW;    // re, im, -im (x 2 clips) + 2 dummy rows (the Re lanes' second write)
    __attribute__((aligned(16))) unsigned char vec[2][VEC_BYTES];      // [parity]: the un-normalised ut (forward) / ybar (reverse)
};
template <int W>
__device__ __forceinline__ float sum_waves(const float* p) {          // p[0..W-1], 16-byte aligned
    if constexpr (W == 4) {
        const f4 t = *reinterpret_cast<const f4*>(p);
        return (t.x + t.y) + (t.z + t.w);
    } else if constexpr (W == 3) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        return (t.x + t.y) + p[2];
    } else {
        const float2 t = *reinterpret_cast<const float2*>(p);
        return t.x + t.y;
    }
}

__device__ __forceinline__ unsigned lds_addr_of(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// ---- the lane geometry of a chain wave ----
// Wave w reads the K-steps in ITS OWN order: local step tau = 0 is the K range this wave itself produced (rows 32 w .. 32 w + 31 of
// the vector: global K-step w of the M_re half, KH + w of the M_im half), tau = 1 .. KH - 1 follow cyclically.  The fragments are
// loaded in that order, so the step code is the same for all waves, and the two own K-steps need no barrier: they are read back
// right behind the wave's own stores and their eight MFMAs start the moment the barrier opens, while the reads of the other waves'
// ranges are still in flight (rounds 1-3 and the first 16x16x32 version: ~110 cycles of LDS latency with an idle matrix pipe).
template <int KH>
struct ChainLane {
    int j, f, q, ia, ib;          // B column / D lane, form, clip, the two adjacent rows
    bool odd;                     // this lane's component: Im (odd) or Re
    unsigned lo[KH], hi[KH];      // LDS byte addresses (parity 0 image) of the A operand of local K-step tau: K < D (array re / im) and K >= D (-im / re)
    int wr1, wr2;                 // byte offsets of the rows this lane writes: own array; -im (Im lanes) or the dummy row (Re lanes)
};
// STACK (the chain16 kernels, round 5): A rows 4 f + 2 / 4 f + 3 carry the LO fp16 piece of the own / partner form (read from the piece-1
// image, VEC bytes behind the piece-0 image) instead of repeating row 4 f: one MFMA then multiplies BOTH pieces of the vector with a
// matrix piece (registers 0, 1: hi piece; 2, 3: lo piece; the caller adds them), one LDS read per K-step feeds it.
template <int PD, bool STACK = false>
__device__ __forceinline__ ChainLane<PD / 32> chain_lane(int w, int lane, unsigned img0) {
    constexpr int VROW = PairLds<PD>::VROW, KH = PD / 32;
    ChainLane<KH> g;
    g.j = lane & 15; g.f = lane >> 4; g.q = g.f >> 1; g.odd = (g.f & 1) != 0;
    g.ia = 32 * w + 2 * g.j; g.ib = g.ia + 1;
    // A rows 4 f (register 0 of the result: this lane's own form) and 4 f + 1 (register 1: the Re <-> Im PARTNER's form, so that no
    // lane ever has to fetch its partner's value from 16 lanes away); the other rows repeat row 4 f (their results are not used)
    const int af = ((lane >> 2) & 3) ^ ((STACK ? (lane & 1) == 1 : (lane & 3) == 1) ? 1 : 0), aq = af >> 1, kg = lane >> 4;
    const bool ac1 = (af & 1) != 0;                                   // c1 = [u_im; u_re], c0 = [u_re; -u_im]
    // The piece-1 image lies VEC = 8 VROW bytes (a multiple of the 256-B bank row) behind piece 0, so the same array of both pieces falls on
    // the same banks.  A 16-lane group of the operand's ds_read_b128 touches sixteen different 16-byte pieces now -- (own, partner) x (piece
    // 0, 1) x four (form, K group) pairs -- and has exactly sixteen slots: the piece-1 image therefore stores array KIND k (re | im | -im |
    // dummy) in the place of kind 3 - k, which puts its four arrays of either K half on the bank slots the piece-0 reads leave free
    // (measured before this: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.38 / 0.31 in the two chain kernels, every operand read 2-way).
    const bool p1 = STACK && (lane & 2);
    const unsigned pc1 = p1 ? 8u * VROW : 0u;
    const int kind_lo = ac1 ? 1 : 0, kind_hi = ac1 ? 0 : 2;
    const unsigned rd_lo = img0 + pc1 + ((p1 ? 3 - kind_lo : kind_lo) * 2 + aq) * VROW + 16 * kg;
    const unsigned rd_hi = img0 + pc1 + ((p1 ? 3 - kind_hi : kind_hi) * 2 + aq) * VROW + 16 * kg;
#pragma unroll
    for (int t = 0; t < KH; ++t) {
        g.lo[t] = rd_lo + 64 * ((t + w) % KH);
        g.hi[t] = rd_hi + 64 * ((t + w) % KH);
    }
    g.wr1 = ((g.odd ? 1 : 0) * 2 + g.q) * VROW + g.ia * 2;
    g.wr2 = ((g.odd ? 2 : 3) * 2 + g.q) * VROW + g.ia * 2;
    return g;
}
// rows ia, ia + 1 are adjacent: one packed 4-byte store per array
template <int KH>
__device__ __forceinline__ void write_vec(unsigned char* base, const ChainLane<KH>& g, float xa, float xb) {
    const unsigned pk = pk_bf16(xa, xb);
    *reinterpret_cast<unsigned*>(base + g.wr1) = pk;
    *reinterpret_cast<unsigned*>(base + g.wr2) = pk ^ 0x80008000u;
}

// B fragments of one matrix for this lane in the wave's K-step order: frag[tile * KS + tau] (M_re half) and frag[tile * KS + KH + tau]
// (M_im half) = 8 bf16 = columns 32 ((tau + w) % KH) + 8 kg .. + 7 of that half in row ia + tile; elem(tile, half, col) is the float32 entry
template <int PD, bool AGPR, typename F>
__device__ __forceinline__ void load_frags(u4 (&frag)[PD / 8], int w, int kg, F&& elem) {
    constexpr int KS = PD / 16, KH = PD / 32;
#pragma unroll
    for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            unsigned v[4];
            const int half = t / KH, col0 = 32 * ((t % KH + w) % KH) + 8 * kg;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                v[e] = (unsigned)bf16_rne(elem(tile, half, col0 + 2 * e)) | ((unsigned)bf16_rne(elem(tile, half, col0 + 2 * e + 1)) << 16);
            frag[tile * KS + t] = u4{v[0], v[1], v[2], v[3]};
            // into its registers now: the loads of all fragments in flight at once would be the kernel's register peak
            if constexpr (AGPR) asm volatile("" : "+a"(frag[tile * KS + t]));
            else asm volatile("" : "+v"(frag[tile * KS + t]));
        }
}

#if defined(CMPS_DIAG) && (defined(PABL_NO_MFMA) || defined(PABL_LOSS_NO_MFMA))      // diagnostic builds only (scripts/ablate.py)
#define PAIR_LASM(TEXT) "s_nop 0"
#else
#define PAIR_LASM(TEXT) TEXT
#endif
// ---- LDS reads of the A operands, all in asm: issued back to back, awaited K-step by K-step with counted lgkmcnt waits (LDS
// operations of a wave complete in order, so whatever else is in flight only makes a counted wait stricter) ----
// the wave's own K-steps (tau = 0), read back right behind its own stores of image parity `POFF / stride`
template <int POFF>
__device__ __forceinline__ void rd_own(unsigned lo0, unsigned hi0, u4& vlo, u4& vhi) {
    asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"
                 : "=&v"(vlo), "=&v"(vhi) : "v"(lo0), "v"(hi0), "n"(POFF) : "memory");
}
// two 16-byte table rows (x0, x1: per-step scalars, rho rows, norm partials) and then the K-steps tau = 1 .. KH - 1 of both halves:
// 2 + 2 (KH - 1) reads; v[tau - 1] / v[KH - 1 + tau - 1]
template <int KH, int POFF, bool TABLES_LAST>
__device__ __forceinline__ void rd_rest(unsigned ax0, unsigned ax1, const unsigned (&lo)[KH], const unsigned (&hi)[KH], f4& x0, f4& x1,
                                        u4 (&v)[2 * KH - 2]) {
#if defined(CMPS_DIAG) && defined(PABL_NO_READS)      // diagnostic builds only (scripts/ablate.py): the table rows only
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(x0), "=&v"(x1) : "v"(ax0), "v"(ax1) : "memory");
    for (int i = 0; i < 2 * KH - 2; ++i) { v[i] = u4{lo[0], hi[0], ax0, ax1}; asm volatile("" : "+v"(v[i])); }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return;
#endif
    if constexpr (!TABLES_LAST) {
        if constexpr (KH == 4)
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %9\n\t"
                         "ds_read_b128 %2, %10 offset:%16\n\tds_read_b128 %3, %11 offset:%16\n\tds_read_b128 %4, %12 offset:%16\n\t"
                         "ds_read_b128 %5, %13 offset:%16\n\tds_read_b128 %6, %14 offset:%16\n\tds_read_b128 %7, %15 offset:%16"
                         : "=&v"(x0), "=&v"(x1), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                         : "v"(ax0), "v"(ax1), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "n"(POFF) : "memory");
        else if constexpr (KH == 3)
            asm volatile("ds_read_b128 %0, %6\n\tds_read_b128 %1, %7\n\t"
                         "ds_read_b128 %2, %8 offset:%12\n\tds_read_b128 %3, %9 offset:%12\n\t"
                         "ds_read_b128 %4, %10 offset:%12\n\tds_read_b128 %5, %11 offset:%12"
                         : "=&v"(x0), "=&v"(x1), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                         : "v"(ax0), "v"(ax1), "v"(lo[1]), "v"(lo[2]), "v"(hi[1]), "v"(hi[2]), "n"(POFF) : "memory");
        else
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\t"
                         "ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8"
                         : "=&v"(x0), "=&v"(x1), "=&v"(v[0]), "=&v"(v[1])
                         : "v"(ax0), "v"(ax1), "v"(lo[1]), "v"(hi[1]), "n"(POFF) : "memory");
    } else {                       // the operands first: the table rows are not needed before the step's tail
        if constexpr (KH == 4)
            asm volatile("ds_read_b128 %2, %10 offset:%16\n\tds_read_b128 %3, %11 offset:%16\n\tds_read_b128 %4, %12 offset:%16\n\t"
                         "ds_read_b128 %5, %13 offset:%16\n\tds_read_b128 %6, %14 offset:%16\n\tds_read_b128 %7, %15 offset:%16\n\t"
                         "ds_read_b128 %0, %8\n\tds_read_b128 %1, %9"
                         : "=&v"(x0), "=&v"(x1), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5])
                         : "v"(ax0), "v"(ax1), "v"(lo[1]), "v"(lo[2]), "v"(lo[3]), "v"(hi[1]), "v"(hi[2]), "v"(hi[3]), "n"(POFF) : "memory");
        else if constexpr (KH == 3)
            asm volatile("ds_read_b128 %2, %8 offset:%12\n\tds_read_b128 %3, %9 offset:%12\n\t"
                         "ds_read_b128 %4, %10 offset:%12\n\tds_read_b128 %5, %11 offset:%12\n\t"
                         "ds_read_b128 %0, %6\n\tds_read_b128 %1, %7"
                         : "=&v"(x0), "=&v"(x1), "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
                         : "v"(ax0), "v"(ax1), "v"(lo[1]), "v"(lo[2]), "v"(hi[1]), "v"(hi[2]), "n"(POFF) : "memory");
        else
            asm volatile("ds_read_b128 %2, %6 offset:%8\n\tds_read_b128 %3, %7 offset:%8\n\t"
                         "ds_read_b128 %0, %4\n\tds_read_b128 %1, %5"
                         : "=&v"(x0), "=&v"(x1), "=&v"(v[0]), "=&v"(v[1])
                         : "v"(ax0), "v"(ax1), "v"(lo[1]), "v"(hi[1]), "n"(POFF) : "memory");
    }
}
// wait until at most W LDS operations are outstanding; the named registers are the reads this makes available
template <int W>
__device__ __forceinline__ void lds_wait2(f4& a, f4& b) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(W) : "memory");
}
// One K-step: the A operand v against the fragments of two matrices x two tiles.  The MFMAs are compiler builtins: hipcc then places
// the independent VALU work of a step BETWEEN them (about two instructions behind each: what a lone wave can hide -- eight in a
// block behind four MFMAs cost it 46 % more, scripts/ubench/mfma16_matvec.hip) and pads the accumulator hazards itself (rounds 1-3
// and the first 16x16x32 version wrote them as asm statements, which the scheduler treats as opaque blocks).  What the compiler
// cannot see is when the operand's LDS read (issued in asm) has landed: the counted wait is an asm statement that "modifies" the
// operand, and a scheduling barrier behind every K-step keeps the waits from being gathered in front of the first MFMA.
// Where the fragments live is the kernel's choice: the forward (two waves per SIMD, 256 registers each) has no AGPR operand
// anywhere, so hipcc selects the VGPR form of the MFMA and all 256 registers are one file (with "a" operands in the kernel it
// selects the AGPR form, whose accumulators the tail has to read back with v_accvgpr_read, and the fixed 128 / 128 split spilled);
// the reverse scan (one wave per SIMD, 512 registers) pins them in AGPRs (load_frags<.., true>: gfx950 MFMAs read A / B operands
// from either file) and reads the eight accumulator values it needs back once per step.
typedef short bf8 __attribute__((ext_vector_type(8)));
template <int W, bool FIRST>
__device__ __forceinline__ void kstep(const u4& fa0, const u4& fa1, const u4& fb0, const u4& fb1, u4& v, f4& a0, f4& a1, f4& b0, f4& b1) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(W) : "memory");
    if constexpr (FIRST) { a0 = f4{0.f, 0.f, 0.f, 0.f}; a1 = a0; b0 = a0; b1 = a0; }
    const bf8 av = __builtin_bit_cast(bf8, v);
#if defined(CMPS_DIAG) && defined(PABL_NO_MFMA)       // diagnostic builds only (scripts/ablate.py)
    a0[0] += __uint_as_float(v.x); a1[0] += __uint_as_float(fa1.x); b0[0] += __uint_as_float(fb0.x); b1[0] += __uint_as_float(fb1.x + fa0.x);
#else
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fa0), a0, 0, 0, 0);
    b0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fb0), b0, 0, 0, 0);
#if defined(CMPS_DIAG) && defined(PABL_HALF_MFMA)     // diagnostic builds only: tile 0 alone (what an MFMA costs in place: the difference, / 16)
    a1[0] += __uint_as_float(fa1.x); b1[0] += __uint_as_float(fb1.x);
#else
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fa1), a1, 0, 0, 0);
    b1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fb1), b1, 0, 0, 0);
#endif
#endif
}
// A mat-vec pair after the barrier: a0 / a1 = (A v) rows ia / ia + 1, b0 / b1 = (B v); register 0 = this lane's own form, register 1
// = its partner's.  Order: the table rows and the other waves' K ranges are requested (rd_rest), the two OWN K-steps (already in
// registers: vlo, vhi) go to the matrix pipe at once, piece(ic<0>) with the two table rows, then K-step i = 1 .. 2 KH - 2 of the
// rest with piece(ic<i>) behind it (64 matrix-pipe cycles each, of which the four MFMAs' own issue takes 32: room for about eight
// independent VALU instructions), ic<2 KH - 1 .. 7> behind the last.  The compiler does not know what the asm statements cost and
// schedules plain code around them freely: a piece that must stay where it is called pins its inputs when it starts and its
// results when it ends (PAIR_PIN).
template <int I> struct ic { static constexpr int value = I; };
#define PAIR_PIN1(a) asm volatile("" : "+v"(a))
#define PAIR_PIN2(a, b) asm volatile("" : "+v"(a), "+v"(b))
#define PAIR_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
// the issue order inside a K-step region: MFMA, two VALU, MFMA, two VALU ... -- what a lone wave hides behind a 16x16x32 MFMA
// (scripts/ubench/mfma16_matvec.hip: one or two plain VALU behind each MFMA are free, a block of eight behind four costs 46 % more);
// left alone the scheduler put a piece in front of its region's MFMAs with the matrix pipe idle.  The file is compiled with
// -fno-slp-vectorize: packed-f32 VALU is expensive beside MFMAs, and the scheduler's packing added a v_mov shuffle per operand.
template <int N>
__device__ __forceinline__ void mfma_valu_pipeline() {
#pragma unroll
    for (int i = 0; i < N; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);     // two VALU
    }
}
template <int PD, int I, int XTRA, typename Piece>
__device__ __forceinline__ void ksteps_rest(const u4 (&FA)[PD / 8], const u4 (&FB)[PD / 8], u4 (&v)[PD / 16 - 2], f4& a0, f4& a1, f4& b0, f4& b1,
                                            Piece&& piece) {
    constexpr int KS = PD / 16, KH = PD / 32, NR = KS - 2;     // NR reads of the rest: lo tau = 1 .. KH - 1, then hi tau = 1 .. KH - 1
    if constexpr (I < NR) {
        constexpr int T = I < KH - 1 ? 1 + I : KH + 1 + (I - (KH - 1));   // index into the wave's fragment order
        kstep<NR - 1 - I + XTRA, false>(FA[T], FA[KS + T], FB[T], FB[KS + T], v[I], a0, a1, b0, b1);
        piece(ic<I + 1>{});
        mfma_valu_pipeline<4>();
        __builtin_amdgcn_sched_barrier(0);
        ksteps_rest<PD, I + 1, XTRA>(FA, FB, v, a0, a1, b0, b1, piece);
    } else if constexpr (I < 7) {
        piece(ic<I + 1>{});
        ksteps_rest<PD, I + 1, XTRA>(FA, FB, v, a0, a1, b0, b1, piece);
    }
}
// read number IDX of a step's LDS read sequence (the order the counted waits of matvec2 assume): the two table rows first or last,
// the other waves' K ranges (M_re half tau = 1 .. KH - 1, then the M_im half) in between
template <int KH, int POFF, bool TABLES_LAST, int IDX>
__device__ __forceinline__ void rd_seq(unsigned ax0, unsigned ax1, const unsigned (&lo)[KH], const unsigned (&hi)[KH], f4& x0, f4& x1, u4 (&v)[2 * KH - 2]) {
    constexpr int NV = 2 * KH - 2;
    constexpr int vi = TABLES_LAST ? IDX : IDX - 2;             // index into v, or a table row outside [0, NV)
    if constexpr (IDX >= NV + 2) {
    } else if constexpr (vi >= 0 && vi < NV) {
        if constexpr (vi < KH - 1) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v[vi]) : "v"(lo[1 + vi]), "n"(POFF) : "memory");
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v[vi]) : "v"(hi[1 + vi - (KH - 1)]), "n"(POFF) : "memory");
    } else if constexpr ((TABLES_LAST ? IDX - NV : IDX) == 0) {
        asm volatile("ds_read_b128 %0, %1" : "=&v"(x0) : "v"(ax0) : "memory");
    } else {
        asm volatile("ds_read_b128 %0, %1" : "=&v"(x1) : "v"(ax1) : "memory");
    }
}
// an own K-step (its operand arrived before the barrier) with four of the step's LDS reads issued behind its MFMAs, one each: in one
// burst in front of the own K-steps the 2 KH reads took ~25 cycles apiece with the matrix pipe idle (measured on k_fwd_chain16,
// profiles/r4_c5wide_chain16_ablations.log)
template <bool FIRST, typename Rd>
__device__ __forceinline__ void kstep_own(const u4& fa0, const u4& fa1, const u4& fb0, const u4& fb1, u4& v, f4& a0, f4& a1, f4& b0, f4& b1, Rd&& rd) {
    // The operand was read in the previous step's tail (rd_own, asm) and has landed by the barrier's lgkmcnt(0) -- but the compiler only
    // sees a register defined by that asm: without this statement (asm volatile statements keep their order, the barrier is one) it is
    // free to issue the MFMAs that read it ABOVE the barrier.  (The first interleaved version had dropped the no-op wait that did this
    // job in kstep<>: intermittent garbage in tests/test_gpu_pair.py, about every second run.)
    asm volatile("" : "+v"(v) :: "memory");
    if constexpr (FIRST) { a0 = f4{0.f, 0.f, 0.f, 0.f}; a1 = a0; b0 = a0; b1 = a0; }
    const bf8 av = __builtin_bit_cast(bf8, v);
#if defined(CMPS_DIAG) && defined(PABL_NO_MFMA)       // diagnostic builds only (scripts/ablate.py)
    a0[0] += __uint_as_float(v.x); a1[0] += __uint_as_float(fa1.x); b0[0] += __uint_as_float(fb0.x); b1[0] += __uint_as_float(fb1.x + fa0.x);
    rd(ic<0>{}); rd(ic<1>{}); rd(ic<2>{}); rd(ic<3>{});
#else
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fa0), a0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0); rd(ic<0>{}); __builtin_amdgcn_sched_barrier(0);
    b0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fb0), b0, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0); rd(ic<1>{}); __builtin_amdgcn_sched_barrier(0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fa1), a1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0); rd(ic<2>{}); __builtin_amdgcn_sched_barrier(0);
    b1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, __builtin_bit_cast(bf8, fb1), b1, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0); rd(ic<3>{}); __builtin_amdgcn_sched_barrier(0);
#endif
}
template <int PD, int POFF, bool TABLES_LAST, typename Piece>
__device__ __forceinline__ void matvec2(const u4 (&FA)[PD / 8], const u4 (&FB)[PD / 8], const ChainLane<PD / 32>& g, unsigned ax0, unsigned ax1,
                                        u4& vlo, u4& vhi, f4& x0, f4& x1, f4& a0, f4& a1, f4& b0, f4& b1, Piece&& piece) {
    constexpr int KS = PD / 16, KH = PD / 32;
    u4 v[KS - 2];
#if defined(CMPS_DIAG) && (defined(PABL_NO_READS) || defined(PABL_READ_BURST))     // diagnostic / A/B builds: all reads in front of the own K-steps (rounds 1-3 and the first round-4 form)
    rd_rest<KH, POFF, TABLES_LAST>(ax0, ax1, g.lo, g.hi, x0, x1, v);
    __builtin_amdgcn_sched_barrier(0);
    kstep<KS, true>(FA[0], FA[KS], FB[0], FB[KS], vlo, a0, a1, b0, b1);          // (the counts are no-ops: vlo, vhi arrived
    kstep<KS, false>(FA[KH], FA[KS + KH], FB[KH], FB[KS + KH], vhi, a0, a1, b0, b1);   // before the barrier)
#else
    kstep_own<true>(FA[0], FA[KS], FB[0], FB[KS], vlo, a0, a1, b0, b1,
                    [&](auto i_) { rd_seq<KH, POFF, TABLES_LAST, decltype(i_)::value>(ax0, ax1, g.lo, g.hi, x0, x1, v); });
    kstep_own<false>(FA[KH], FA[KS + KH], FB[KH], FB[KS + KH], vhi, a0, a1, b0, b1,
                     [&](auto i_) { rd_seq<KH, POFF, TABLES_LAST, 4 + decltype(i_)::value>(ax0, ax1, g.lo, g.hi, x0, x1, v); });
#endif
    if constexpr (!TABLES_LAST) lds_wait2<KS - 2>(x0, x1);
    piece(ic<0>{});
    mfma_valu_pipeline<8>();
    __builtin_amdgcn_sched_barrier(0);
    ksteps_rest<PD, 0, TABLES_LAST ? 2 : 0>(FA, FB, v, a0, a1, b0, b1, piece);
    if constexpr (TABLES_LAST) lds_wait2<0>(x0, x1);
}

// rho rows are staged through LDS one 32-step chunk at a time (a row per step straight from L2 / HBM costs its full
// latency every step: the table is 16 MB at D = 128); the next chunk is loaded into registers at the start of a chunk and
// committed to the other buffer in the middle of it
constexpr int RCH = 32;
template <int D>
struct RhoStage {
    __attribute__((aligned(16))) float2 row[2][RCH][D];
};
// D / 32 waves (tid < 2 D) move one chunk: 32 rows x D / 2 float4 = 8 float4 per thread.  Eight named values, not an array
// (the array went through scratch memory); all loads are issued before the first store.
template <int D>
__device__ __forceinline__ void rho_stage(const Dev& P, RhoStage<D>& S, int chunk, int buf, int tid) {
    const float4* src = reinterpret_cast<const float4*>(P.rho);       // [N + 1][D] float2 = D / 2 float4 per row
    const int maxrow = P.N;                                            // the table has N + 1 rows
    auto ld = [&](int i) {
        const int e = tid + 2 * D * i;
        int rowi = chunk * RCH + e / (D / 2);
        rowi = rowi < 0 ? 0 : (rowi > maxrow ? maxrow : rowi);
        return src[(size_t)rowi * (D / 2) + e % (D / 2)];
    };
    const float4 r0 = ld(0), r1 = ld(1), r2 = ld(2), r3 = ld(3), r4 = ld(4), r5 = ld(5), r6 = ld(6), r7 = ld(7);
    __builtin_amdgcn_sched_barrier(0);
    float4* dst = reinterpret_cast<float4*>(&S.row[buf][0][0]);
    dst[tid] = r0;             dst[tid + 2 * D] = r1;      dst[tid + 4 * D] = r2;      dst[tid + 6 * D] = r3;
    dst[tid + 8 * D] = r4;     dst[tid + 10 * D] = r5;     dst[tid + 12 * D] = r6;     dst[tid + 14 * D] = r7;
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// forward: 2 D / 32 waves per workgroup.  The first D / 32 ("chain") carry the recurrence u -> y -> u' with the R and Q fragments
// and meet at ONE LDS-only barrier per step.  The others ("loss") do everything nothing waits for -- H y, e = Re(y^dagger H y), the
// stash rows, the per-step scalars and the loss -- and do it EIGHT STEPS AT A TIME: the chain leaves every y_k in an LDS
// ring (bf16 images for the matrix cores, float32 for the dot product), and  H [y_k .. y_{k+7}]  of both clips is one
// 32 x 32 tile of v_mfma_f32_32x32x16_bf16 per loss wave (columns = (component, step, clip); 2 D / 16 MFMAs per batch, spread
// two per step over the following batch so that no step's barrier waits for them).  Per step a loss wave issues about a
// dozen instructions, and both kinds share a SIMD's issue slots and matrix pipe.
// Stash (Dev::stash, pair layout): per pair and step  [y | H y][clip][re | im][D] float32.
// ------------------------------------------------------------------------------------------------
namespace {

constexpr int FB = 8;                                   // steps per loss-wave batch
// Diagnostic knob (-DPAIR_LOSS_SLEEP=n): a loss wave sleeps ~64 n cycles behind every barrier, which moves its LDS reads behind the chain
// waves' operand burst and its two MFMAs into the chain's tail.  Measured at C5: 0 / 3 / 6 / 9 -> 9.67 / 9.91 / 10.17 / 10.46 ms per
// forward scan (profiles/r4_c5_ablations.log): what matters is that the loss waves reach the next barrier early, so the default is 0.
#ifndef PAIR_LOSS_SLEEP
#define PAIR_LOSS_SLEEP 0
#endif
constexpr int LOSS_SLEEP = PAIR_LOSS_SLEEP;
template <int D>
struct FwdRing {
    static constexpr int BROW = D * 2 + 16, FROW = D * 4 + 16;         // padded rows (bank spread for the loss waves' reads)
    // [slot][row][BROW] bf16 + 64 B per slot.  Row of (clip, kind = re | im | -im | dummy): brow().  A 16-lane group of the loss waves'
    // ds_read_b128 holds sixteen (step, clip, component) columns; with the rows in [clip][kind] order and slots 8 BROW apart they fell on
    // eight 16-byte bank slots (every read 2-way: the kernel's SQ_LDS_BANK_CONFLICT fraction 0.15).  Clip 1's rows reversed and 64 B between
    // slots: sixteen different slots for either operand half (brute-forced over row orders and paddings, the chain waves' 4-byte stores
    // keep their 2-way overlap: there is no layout without either).
    static constexpr int BSLOT = 8 * BROW + 64;
    static __device__ __forceinline__ constexpr int brow(int clip, int kind) { return clip ? 7 - kind : kind; }
    __attribute__((aligned(16))) unsigned char b[2 * FB][BSLOT];
    __attribute__((aligned(16))) unsigned char f[2 * FB][2][2][FROW];  // [slot][clip][re | im] float32
    __attribute__((aligned(16))) float nrm[2 * FB][2][4];              // [slot][clip][chain wave]: partial |y|^2
    __attribute__((aligned(16))) float ee[2][2 * FB][4];               // [batch parity][step in batch * 2 + clip][loss wave]
    __attribute__((aligned(16))) float sinc[2][2][PCH];                // [chunk parity][clip][step]: s_k = x_k / A of a 64-step chunk (loss wave 0 -> chain waves)
};

typedef float f16t __attribute__((ext_vector_type(16)));

// float index of (pair, step, y / H y, clip, component, row) in the pair stash
template <int PD>
__device__ __forceinline__ size_t pair_stash_index(size_t pair, int N, int step, int yh, int clip, int comp, int row) {
    return (((((pair * N + step) * 2 + yh) * 2 + clip) * 2 + comp) * PD) + row;
}

}  // namespace

template <int PD, bool SAVE>
__global__ __launch_bounds__(4 * PD, 1) void k_fwd_pair(Dev P, const float* __restrict__ audio,
                                                        float* __restrict__ loss_out) {
    constexpr int PWV = PD / 32;
    constexpr int BROW = FwdRing<PD>::BROW, FROW = FwdRing<PD>::FROW;
    __shared__ PairLds<PD> L;
    __shared__ RhoStage<PD> RS;
    __shared__ FwdRing<PD> RG;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int w = wv % PWV;
    const bool loss_wave = wv >= PWV;
    const int N = P.N, T = P.T, NC = (N + PCH - 1) / PCH;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;   // an odd batch repeats its last clip (not stored)
    const bool two = b1 != b0;
    const float* xr0 = audio + (size_t)b0 * T;
    const float* xr1 = audio + (size_t)b1 * T;
    const float A = dev_A(P);
    // both kinds run the same number of iterations (one barrier each): batch bt of the loss waves multiplies batch bt - 1 and
    // finishes batch bt - 2
    const int NBT = (N - 1) / FB + 3;

    if (!loss_wave) {
        // ================================================================== chain waves
        __builtin_amdgcn_s_setprio(3);      // both kinds share a SIMD's matrix pipe: the serial chain goes first
        constexpr int KH = PD / 32, VEC = PairLds<PD>::VEC_BYTES;
        u4 FR[PD / 8], FQ[PD / 8];
        {
            const int row0 = 32 * w + 2 * (lane & 15);                 // rows ia (tile 0) and ia + 1 (tile 1)
            const float2* Rrow = P.R + (size_t)row0 * PD;
            const float2* Qrow = P.Q + (size_t)row0 * PD;
            load_frags<PD, false>(FR, w, lane >> 4, [&](int tile, int half, int c) { return half ? Rrow[tile * PD + c].y : Rrow[tile * PD + c].x; });
            load_frags<PD, false>(FQ, w, lane >> 4, [&](int tile, int half, int c) { return half ? Qrow[tile * PD + c].y : Qrow[tile * PD + c].x; });
        }
        // (everything below is derived from a laundered copy of the lane number: addresses computed before the fragment loads would
        // be live across their register peak, and the allocator then keeps them in scratch memory for the whole kernel)
        int lane_c = lane;
        asm volatile("" : "+v"(lane_c));
        const ChainLane<KH> g = chain_lane<PD>(w, lane_c, lds_addr_of(&L.vec[0][0]));
        const int q = g.q, ia = g.ia, ib = g.ib;
        const bool odd = g.odd;
        const float sg = odd ? 1.f : -1.f;                             // (rho y)_own = rho_re y_own + sg rho_im y_partner
        // the ring rows of this lane within a slot: bf16 (own array, and -im / dummy), float32
        const int rb1 = FwdRing<PD>::brow(q, odd ? 1 : 0) * BROW + ia * 2, rb2 = FwdRing<PD>::brow(q, odd ? 2 : 3) * BROW + ia * 2;
        const int rf = (q * 2 + (odd ? 1 : 0)) * FROW + ia * 4;
        const unsigned a_nrm = lds_addr_of(&RG.nrm[0][q][0]);        // + 32 slot
        const unsigned a_rho = lds_addr_of(&RS.row[0][0][ia]);        // + 8 PD (32 buffer + row)
        const float2 pa = P.psi0[ia], pb = P.psi0[ib];
        // ut_0 = psi_0 (both clips): (own component, a copy of the partner's -- kept in step by the MFMAs' register 1)
        f2 ua = odd ? f2{pa.y, pa.x} : f2{pa.x, pa.y}, ub = odd ? f2{pb.y, pb.x} : f2{pb.x, pb.y};
        float sv0 = 0.f, sv1 = 0.f;                                   // s = x / A of the current 64 steps, lane <-> step
        u4 vlo, vhi;                                                  // the wave's own K-steps of the image the next step multiplies
        write_vec(L.vec[0], g, ua.x, ub.x);
        rd_own<0>(g.lo[0], g.hi[0], vlo, vhi);
        rho_stage<PD>(P, RS, 0, 0, 64 * w + lane_c);
        __syncthreads();
#if defined(CMPS_DIAG) && defined(PABL_TIMING)        // diagnostic builds only: where a chain step's cycles go (s_memtime stamps, consumed a step late)
        unsigned long long fA = 0, fB = 0, fC = 0, fA1 = 0, fB1 = 0, fC1 = 0, fC2 = 0, fPre = 0, fMv = 0, fTail = 0, fN = 0;
#define PAIR_FSTAMP_A() { if (fC2) { fPre += fA1 - fC2; fMv += fB1 - fA1; fTail += fC1 - fB1; ++fN; } fC2 = fC1; fA = __builtin_readcyclecounter(); }
#define PAIR_FSTAMP_B() fB = __builtin_readcyclecounter()
#define PAIR_FSTAMP_C(x) { float x_ = x; asm volatile("" : "+v"(x_)); fC = __builtin_readcyclecounter(); fA1 = fA; fB1 = fB; fC1 = fC; }
#else
#define PAIR_FSTAMP_A()
#define PAIR_FSTAMP_B()
#define PAIR_FSTAMP_C(x)
#endif
        // Eight steps per block with the step-in-block J static: the LDS parities and ring slots are immediates and the chunk
        // boundaries (increments every 64 steps, rho staging every 32) can only fall on J = 0.
        // A step: [barrier] -> table rows + the other waves' K ranges requested, own K-steps to the matrix pipe, the rest as it arrives
        // -> y_k (own and partner copy, packed float32) -> ut_{k+1} -> bf16 image, own K ranges read back -> ring rows, |y|^2 -> [barrier]
#define PAIR_FWD_STEP(J)                                                                                                   \
        {                                                                                                                  \
            constexpr int p = (J) & 1;                                                                                     \
            const int k = FB * bt + (J);                                                                                   \
            if ((J) == 0 && (bt & (PCH / FB - 1)) == 0) {              /* increments of the next 64 steps, one per lane */  \
                if (bt == 0) {                                                                                             \
                    const int idx = k + lane_c;                                                                            \
                    const bool in0 = idx < T, in1 = idx + 1 < T;                                                           \
                    sv0 = ((in1 ? xr0[idx + 1] : 0.f) - (in0 ? xr0[idx] : 0.f)) / A;     /* model.py:263, 303 */            \
                    sv1 = ((in1 ? xr1[idx + 1] : 0.f) - (in0 ? xr1[idx] : 0.f)) / A;                                       \
                } else {                                               /* left in LDS a chunk ago by loss wave 0 (below) */ \
                    sv0 = RG.sinc[(bt / (PCH / FB)) & 1][0][lane_c];                                                       \
                    sv1 = RG.sinc[(bt / (PCH / FB)) & 1][1][lane_c];                                                       \
                }                                                                                                          \
            }                                                                                                              \
            const int hb_ = bt & 1;                                    /* the ring half of this block */                   \
            /* |y_{k-1}|^2 partials (published by the previous step; nothing at k = 0) and rho_k of this lane's rows */     \
            const int nslot = (J) > 0 ? hb_ * FB + (J) - 1 : (hb_ ^ 1) * FB + FB - 1;                                      \
            const unsigned ax0 = a_nrm + 32 * nslot;                                                                       \
            const unsigned ax1 = a_rho + 8 * PD * (((bt / (RCH / FB)) & 1) * RCH + (bt & (RCH / FB - 1)) * FB + (J));       \
            f4 xn, rh, cR0, cR1, cQ0, cQ1;                                                                                 \
            float inv, s;                                                                                                  \
            PAIR_FSTAMP_A();                                                                                               \
            matvec2<PD, p * VEC, false>(FR, FQ, g, ax0, ax1, vlo, vhi, xn, rh, cR0, cR1, cQ0, cQ1, [&](auto pc) {                 \
                if constexpr (decltype(pc)::value == 0) {                                                                  \
                    inv = __builtin_amdgcn_rsqf(fmaxf(sum4<PWV>(xn), 1e-12f));           /* model.py:332 */                \
                    if ((J) == 0 && bt == 0) inv = 1.f;                                                                    \
                    const int kl = (bt & (PCH / FB - 1)) * FB + (J);                                                       \
                    const float s0 = rdl(sv0, kl), s1 = rdl(sv1, kl);  /* (both read: a readlane inside a select becomes a branch) */ \
                    s = q ? s1 : s0;                                                                                       \
                    PAIR_PIN2(inv, s);                                                                                     \
                }                                                                                                          \
            });                                                                                                            \
            PAIR_FSTAMP_B();                                                                                               \
            /* y_k, rows ia / ib: own component (register 0) and the partner's (register 1), as explicit (own, partner) pairs: the tail  \
               runs when the matrix pipe is idle, where packed float32 VALU halves its instruction count */                        \
            const f2 ya = inv * (ua + (f2{cQ0[0], cQ0[1]} + s * f2{cR0[0], cR0[1]}));                                       \
            const f2 yb = inv * (ub + (f2{cQ1[0], cQ1[1]} + s * f2{cR1[0], cR1[1]}));                                       \
            const float yna = ya.x, ynb = yb.x;                                                                            \
            /* Two chains start at y and both end in an LDS store the barrier waits for: ut_{k+1} = rho_k y_k -> bf16 image, and    \
               |y_k|^2 -> four dependent DPP adds -> norm partial.  Written out turn by turn (scheduling barriers), the second      \
               runs in the first one's dependency gaps; left to the scheduler it came behind it. */                                \
            const f2 n2 = ya * ya + yb * yb;                                                                               \
            float nn = n2.x + n2.y;                                                                                        \
            const f2 ta = f2{sg * rh.y, -(sg * rh.y)} * __builtin_shufflevector(ya, ya, 1, 0);                              \
            const f2 tb = f2{sg * rh.w, -(sg * rh.w)} * __builtin_shufflevector(yb, yb, 1, 0);                              \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            nn += dpp_mov<0x128>(nn);                                  /* row_ror:8 */                                     \
            ua = rh.x * ya + ta;   ub = rh.z * yb + tb;               /* ut_{k+1} = rho_k y_k (un-normalised), own and partner */ \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            nn += dpp_mov<0x124>(nn);                                  /* row_ror:4 */                                     \
            const unsigned pku = pk_bf16(ua.x, ub.x);                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            nn += dpp_mov<0x122>(nn);                                  /* row_ror:2 */                                     \
            *reinterpret_cast<unsigned*>(L.vec[p ^ 1] + g.wr1) = pku;                                                      \
            *reinterpret_cast<unsigned*>(L.vec[p ^ 1] + g.wr2) = pku ^ 0x80008000u;                                        \
            rd_own<(p ^ 1) * VEC>(g.lo[0], g.hi[0], vlo, vhi);         /* (same wave, in order: no wait between store and read) */ \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            nn += dpp_mov<0x121>(nn);                                  /* row_ror:1: the clip's total over this wave's rows */ \
            const unsigned pky = pk_bf16(yna, ynb);                                                                        \
            __builtin_amdgcn_sched_barrier(0);                                                                             \
            if (lane_c == 0 || lane_c == 32) RG.nrm[hb_ * FB + (J)][q][w] = nn;                                           \
            {   /* y_k for the loss waves: bf16 images and float32 */                                                      \
                unsigned char* rbase = &RG.b[hb_ * FB + (J)][0];                                                           \
                *reinterpret_cast<unsigned*>(rbase + rb1) = pky;                                                           \
                *reinterpret_cast<unsigned*>(rbase + rb2) = pky ^ 0x80008000u;                                             \
                *reinterpret_cast<float2*>(&RG.f[hb_ * FB + (J)][0][0][0] + rf) = make_float2(yna, ynb);                   \
            }                                                                                                              \
            PAIR_FSTAMP_C(nn);                                                                                             \
            lds_barrier();                                                                                                 \
        }
        // ONE straight-line copy of the eight steps, run for every iteration: the steps behind the clip's end (at most 7, and the
        // 16 of the loss waves' run-out) compute on whatever is there and write ring slots the loss waves never look at (they check
        // every step against N themselves).  A second, conditional copy of the block gave the fragments' "a" operands a second
        // register assignment and the allocator moved them through scratch memory and sixteen v_accvgpr copies per K-step.
        for (int bt = 0; bt < NBT; ++bt) {
            PAIR_FWD_STEP(0) PAIR_FWD_STEP(1) PAIR_FWD_STEP(2) PAIR_FWD_STEP(3)
            PAIR_FWD_STEP(4) PAIR_FWD_STEP(5) PAIR_FWD_STEP(6) PAIR_FWD_STEP(7)
        }
#undef PAIR_FWD_STEP
#if defined(CMPS_DIAG) && defined(PABL_TIMING)
        if (blockIdx.x == 0 && threadIdx.x == 0)
            printf("forward chain wave, cycles per step: tail end -> barrier exit %.1f, reads + MFMAs %.1f, tail %.1f\n",
                   (double)fPre / fN, (double)fMv / fN, (double)fTail / fN);
#endif
#undef PAIR_FSTAMP_A
#undef PAIR_FSTAMP_B
#undef PAIR_FSTAMP_C
        return;
    }

    // ====================================================================== loss waves
    // 32x32x16 MFMA views: A row / B column = lane & 31, K half = lane >> 5; D: column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 hk
    const int n = lane & 31, hk = lane >> 5;
    const int comp = n >> 4, sb = (n >> 1) & (FB - 1), clip = n & 1;   // the column of this lane: (component, step in batch, clip)
    constexpr int KS = PD / 16, KT = 2 * KS;                            // MFMAs per batch: H_re [y_re | y_im] and H_im [-y_im | y_re]
    u4 FHre[KS], FHim[KS];                                             // H = R + R^dagger, rows 32 w + n, K = 16 t + 8 hk ..
    {
        const float2* Rrow = P.R + (size_t)(32 * w + n) * PD;
        const float2* RTrow = P.RT + (size_t)(32 * w + n) * PD;       // RT[i][j] = R[j][i]
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            unsigned rr[4], ii[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j0 = 16 * t + 8 * hk + 2 * e;
                rr[e] = (unsigned)bf16_rne(Rrow[j0].x + RTrow[j0].x) | ((unsigned)bf16_rne(Rrow[j0 + 1].x + RTrow[j0 + 1].x) << 16);
                ii[e] = (unsigned)bf16_rne(Rrow[j0].y - RTrow[j0].y) | ((unsigned)bf16_rne(Rrow[j0 + 1].y - RTrow[j0 + 1].y) << 16);
            }
            FHre[t] = u4{rr[0], rr[1], rr[2], rr[3]};
            FHim[t] = u4{ii[0], ii[1], ii[2], ii[3]};
            // one fragment pair at a time ("memory": the loads of the next pair stay behind this point): with all 64 row loads in
            // flight the prologue was the kernel's register peak, and the allocator then kept fragments in scratch memory for the
            // whole loop (reloaded behind s_waitcnt vmcnt(0) in every step: 2 ms of the round-4 forward until this was found)
            asm volatile("" : "+v"(FHre[t]), "+v"(FHim[t]) :: "memory");
        }
    }
    // byte offsets of this lane inside a ring slot: B operand rows (H_re part: own component; H_im part: -im for Re columns,
    // re for Im columns), the float32 rows of its column
    const int ob1 = FwdRing<PD>::brow(clip, comp) * BROW + 16 * hk, ob2 = FwdRing<PD>::brow(clip, comp ? 0 : 2) * BROW + 16 * hk;
    const int of = (clip * 2 + comp) * FROW + (32 * w + 4 * hk) * 4;
    float* stash = reinterpret_cast<float*>(P.stash);
    float* sc_c = SAVE ? P.scal + ((size_t)(clip ? b1 : b0) * NC) * 128 : nullptr;
    const float* xr_c = clip ? xr1 : xr0;
    const bool clip_live = clip == 0 || two;
    float loss0 = 0.f, loss1 = 0.f;
    f16t acc, accE;                                                    // the tile being accumulated / the finished tile of the batch before
    f4 yv[4];                                                          // y (float32) of this lane's column: rows 32 w + 8 g + 4 hk ..+3
    float epE = 0.f, xp0 = 0.f, xp1 = 0.f;
    __syncthreads();
    // Batch bt of the loop: multiplies batch pb = bt - 1 (two MFMAs per step), reads its float32 y (steps 0-3) and stores it
    // (step 4); finishes batch pe = bt - 2 from the tile copied at the end of the iteration before: H y stores (step 0),
    // e partial -> LDS (step 1), e, loss and the scalar rows (step 5).  Nothing is left for one step to do alone.
    for (int bt = 0; bt < NBT; ++bt) {
#if defined(CMPS_DIAG) && defined(PABL_NO_LOSS)       // diagnostic builds only (scripts/ablate.py): the loss waves only keep the barriers
        for (int jj = 0; jj < FB; ++jj) lds_barrier();
        continue;
#endif
        // the next chunk of rho into the other buffer -- here, not in the chain waves (round 5): the loads are consumed at once, a memory
        // latency every 32 steps that cost the chain 2.4 % (ablation: profiles/r5_c5_ablations.log); the step's barrier publishes the rows
        // 32 steps before the chain reads them
        if ((bt & (RCH / FB - 1)) == 0) rho_stage<PD>(P, RS, (FB * bt) / RCH + 1, ((FB * bt) / RCH + 1) & 1, 64 * w + lane);
        if (w == 0 && (bt & (PCH / FB - 1)) == 0) {                   // and the increments of the next 64-step chunk (the same expressions)
            const int cn = bt / (PCH / FB) + 1, idx = cn * PCH + lane;
            const bool in0 = idx < T, in1 = idx + 1 < T;
            RG.sinc[cn & 1][0][lane] = ((in1 ? xr0[idx + 1] : 0.f) - (in0 ? xr0[idx] : 0.f)) / A;
            RG.sinc[cn & 1][1][lane] = ((in1 ? xr1[idx + 1] : 0.f) - (in0 ? xr1[idx] : 0.f)) / A;
        }
        const int pb = bt - 1, pe = bt - 2;
        const bool mul = pb >= 0 && FB * pb < N;
        const bool fin = pe >= 0 && FB * pe < N;
        const unsigned char* bslot = &RG.b[(pb & 1) * FB + sb][0];
        const unsigned char* fslot = &RG.f[(pb & 1) * FB + sb][0][0][0];
        const int step_b = FB * pb + sb, step_e = FB * pe + sb;       // this lane's step in either batch
#if defined(CMPS_DIAG) && defined(PABL_LOSS_NO_READS)     // diagnostic builds only (scripts/ablate.py)
#define PAIR_LOSS_BV(p) u4{(unsigned)(uintptr_t)(p), 1u, 2u, 3u}
#else
#define PAIR_LOSS_BV(p) (*reinterpret_cast<const u4*>(p))
#endif
#define PAIR_LOSS_STEP(J)                                                                                                  \
        {                                                                                                                  \
            if (fin) {                                                                                                     \
                if (((J) & 1) == 0 && SAVE && step_e < N) {          /* H y of batch pe: ONE 16-byte store per lane and step (J = 0, 2, 4, 6) -- */ \
                    constexpr int g = (J) / 2;                         /* four in one step made that step the slowest of the batch, and every    */ \
                    float* hp = stash + pair_stash_index<PD>(blockIdx.x, N, step_e, 1, clip, comp, 32 * w + 4 * hk);   /* barrier waits for the slowest wave */ \
                    *reinterpret_cast<f4*>(hp + 8 * g) = f4{accE[4 * g], accE[4 * g + 1], accE[4 * g + 2], accE[4 * g + 3]}; \
                }                                                                                                          \
                if ((J) == 1) {                                      /* the two audio samples of this lane's step, four steps before their use */ \
                    const bool in = step_e < N;                                                                            \
                    xp0 = in ? xr_c[step_e] : 0.f;                                                                         \
                    xp1 = (in && step_e + 1 < T) ? xr_c[step_e + 1] : 0.f;                                                 \
                }                                                                                                          \
                if ((J) == 1) {                                                                                            \
                    float ep = half_add(epE, epE);                     /* + the other row half */                          \
                    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(ep), __float_as_uint(ep), false, false); \
                    ep = __uint_as_float(r[0]) + __uint_as_float(r[1]);              /* + the other component */            \
                    if (lane < 2 * FB) RG.ee[pe & 1][lane][w] = ep;                                                        \
                }                                                                                                          \
                if ((J) == 5) {                                      /* e of (step, clip) = lane & 15, the loss, the e row */ \
                    const float e = sum_waves<PWV>(&RG.ee[pe & 1][n & 15][0]);                                            \
                    const bool in = step_e < N;                                                                            \
                    const float lv = in ? -logf(1.0f + (e * (xp1 - xp0)) / A) : 0.f;  /* model.py:294 operation order */    \
                    if (SAVE && w == 0 && lane < 2 * FB && in && clip_live)                                                 \
                        sc_c[(size_t)(step_e / PCH) * 128 + 64 + (step_e & (PCH - 1))] = e;                               \
                    _Pragma("unroll") for (int jj = 0; jj < FB; ++jj) {  /* model.py:279: sequential in time */             \
                        loss0 += rdl(lv, 2 * jj);                                                                          \
                        loss1 += rdl(lv, 2 * jj + 1);                                                                      \
                    }                                                                                                      \
                }                                                                                                          \
            }                                                                                                              \
            if (mul) {                                                                                                     \
                if ((J) < 4) yv[(J) & 3] = *reinterpret_cast<const f4*>(fslot + of + 32 * ((J) & 3));                      \
                if (((J) & 1) == 1 && SAVE && step_b < N) {          /* y of batch pb (the repeated clip of an odd batch too): one store per */ \
                    constexpr int g = (J) / 2;                         /* lane at J = 1, 3, 5, 7 (yv[g] was read at J = g)                     */ \
                    float* yp = stash + pair_stash_index<PD>(blockIdx.x, N, step_b, 0, clip, comp, 32 * w + 4 * hk);       \
                    *reinterpret_cast<f4*>(yp + 8 * g) = yv[g];                                                            \
                }                                                                                                          \
                if ((J) == 4 && SAVE && step_b < N && w == 0 && lane < 2 * FB && clip_live)                                 \
                    sc_c[(size_t)(step_b / PCH) * 128 + (step_b & (PCH - 1))] = sum_waves<PWV>(&RG.nrm[(pb & 1) * FB + sb][clip][0]); \
                u4 bvs[(KT + FB - 1) / FB];                                                                                \
                _Pragma("unroll") for (int t = (J) * KT / FB; t < ((J) + 1) * KT / FB; ++t)                                \
                    bvs[t - (J) * KT / FB] = PAIR_LOSS_BV(bslot + (t < KS ? ob1 : ob2) + 32 * (t < KS ? t : t - KS));      \
                if constexpr (LOSS_SLEEP > 0) {                      /* the MFMAs go to the matrix pipe when the chain is in its tail */ \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
                    __builtin_amdgcn_s_sleep(LOSS_SLEEP);                                                                  \
                }                                                                                                          \
                _Pragma("unroll") for (int t = (J) * KT / FB; t < ((J) + 1) * KT / FB; ++t) {                              \
                    const int tt = t < KS ? t : t - KS;                                                                    \
                    const u4 bv = bvs[t - (J) * KT / FB];                                                                  \
                    /* asm (fixed issue points); no AGPR operand anywhere in this kernel: see kstep */                           \
                    if (t == 0)                                                                                            \
                        asm volatile(PAIR_LASM("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0") : "=&v"(acc) : "v"(FHre[0]), "v"(bv));     \
                    else if (t == KT - 1)       /* + the wait states before the VALU reads the tile (8 passes) */          \
                        asm volatile(PAIR_LASM("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0\n\ts_nop 15\n\ts_nop 7")          \
                                     : "+v"(acc) : "v"(FHim[KS - 1]), "v"(bv));                                            \
                    else if (t < KS)                                                                                       \
                        asm volatile(PAIR_LASM("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0") : "+v"(acc) : "v"(FHre[tt]), "v"(bv));   \
                    else                                                                                                   \
                        asm volatile(PAIR_LASM("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0") : "+v"(acc) : "v"(FHim[tt]), "v"(bv));   \
                }                                                                                                          \
                if ((J) == FB - 1) {                                 /* the tile is complete: e partial, hand the tile over */ \
                    float ep = 0.f;                                                                                        \
                    _Pragma("unroll") for (int g = 0; g < 4; ++g)                                                          \
                        _Pragma("unroll") for (int i = 0; i < 4; ++i) ep = fmaf(yv[g][i], acc[4 * g + i], ep);            \
                    epE = ep;                                                                                              \
                    accE = acc;                                                                                            \
                }                                                                                                          \
            }                                                                                                              \
            lds_barrier();                                                                                                 \
        }
        PAIR_LOSS_STEP(0) PAIR_LOSS_STEP(1) PAIR_LOSS_STEP(2) PAIR_LOSS_STEP(3)
        PAIR_LOSS_STEP(4) PAIR_LOSS_STEP(5) PAIR_LOSS_STEP(6) PAIR_LOSS_STEP(7)
#undef PAIR_LOSS_STEP
#undef PAIR_LOSS_BV
    }
    if (w == 0 && lane == 0) {
        loss_out[b0] = loss0;
        if (two) loss_out[b1] = loss1;
    }
}

hipError_t launch_fwd_pair(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    if (P.DP == 128) {
        if (save) hipLaunchKernelGGL((k_fwd_pair<128, true>), dim3(nb), dim3(512), 0, s, P, audio, loss);
        else hipLaunchKernelGGL((k_fwd_pair<128, false>), dim3(nb), dim3(512), 0, s, P, audio, loss);
    } else if (P.DP == 96) {
        if (save) hipLaunchKernelGGL((k_fwd_pair<96, true>), dim3(nb), dim3(384), 0, s, P, audio, loss);
        else hipLaunchKernelGGL((k_fwd_pair<96, false>), dim3(nb), dim3(384), 0, s, P, audio, loss);
    } else if (P.DP == 64) {
        if (save) hipLaunchKernelGGL((k_fwd_pair<64, true>), dim3(nb), dim3(256), 0, s, P, audio, loss);
        else hipLaunchKernelGGL((k_fwd_pair<64, false>), dim3(nb), dim3(256), 0, s, P, audio, loss);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// float32-faithful forward chain on the matrix cores (round 4): the training forward of the WIDE family (CMPS_VARIANT_WIDE / AUTO
// above D = 32) when CMPS_OPT_WIDE_CHAIN = MFMA.  Lane layout, K order and step structure of k_fwd_pair's chain waves, with every
// operand split into TWO fp16 pieces (round to nearest: hi = f16(x), lo = f16(x - hi), 11 + 1 + 11 + 1 bits) and the three products
// hi hi' + hi lo' + lo hi' on v_mfma_f32_16x16x32_f16, as in k_grad_gemm's F16 mode (cmps_grad_gemm.h).  Round 4: 12 MFMAs per K-step (8
// in the QLITE instance); round 5: both pieces of the vector STACKED in the A rows (kstep_st): 8 (6), one operand read instead of two.  fp16 has 5 exponent bits, so the operands are scaled by powers of two chosen ONCE per launch from guaranteed
// bounds: R and Q each by its largest entry (to 2^15), the broadcast vector ut_k = rho_{k-1} y_{k-1} by 1 + |Q|_F + max|s| |R|_F
// (to 2^13): y_k = (1 + Q + s_k R) ut_k / |ut_k|, so |ut_{k+1}| = |y_k| <= 1 + |Q + s_k R|_2.  Only the small correction (Q + s R) ut
// goes through the split; the identity part and everything after the mat-vec is float32, as in every kernel of the family.
// The kernel is the chain alone (4 waves of 512 registers: the 256 AGPRs hold the R and Q pieces): it stashes y_k (the wide family's
// row layout, cmps_wide.hip) and |y_k|^2; k_hy_wide, k_loss_wide, k_bwd_wide and k_grad_gemm run on those rows unchanged.
// ------------------------------------------------------------------------------------------------
namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// (a, b) -> packed hi pieces and packed lo pieces
__device__ __forceinline__ void split_f16x2(float a, float b, unsigned& hi, unsigned& lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    hi = gg::cvt_pk_f16(a, b);
    const h2 h = __builtin_bit_cast(h2, hi);
    lo = gg::cvt_pk_f16(a - (float)h.x, b - (float)h.y);
}

template <int D>
struct Chain16Lds {
    static constexpr int VROW = PairLds<D>::VROW, VEC = 8 * VROW;
    __attribute__((aligned(16))) unsigned char vec[2][2][VEC];     // [parity][piece]: images of ut (arrays re | im | -im | dummy, x 2 clips)
    __attribute__((aligned(16))) float nrm[2][2][4];               // [parity][clip][wave]: partial |y|^2
    __attribute__((aligned(16))) float red[4][4];                  // prologue reductions
};

// hi / lo fragments of one matrix (see load_frags), entries scaled by `scale`
template <int PD, bool PIN_LO = true, typename F>
__device__ __forceinline__ void load_frags_f16(u4 (&fh)[PD / 8], u4 (&fl)[PD / 8], int w, int kg, float scale, F&& elem) {
    constexpr int KS = PD / 16, KH = PD / 32;
#pragma unroll
    for (int tile = 0; tile < 2; ++tile)
#pragma unroll
        for (int t = 0; t < KS; ++t) {
            unsigned vh[4], vl[4];
            const int half = t / KH, col0 = 32 * ((t % KH + w) % KH) + 8 * kg;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                split_f16x2(elem(tile, half, col0 + 2 * e) * scale, elem(tile, half, col0 + 2 * e + 1) * scale, vh[e], vl[e]);
            fh[tile * KS + t] = u4{vh[0], vh[1], vh[2], vh[3]};
            fl[tile * KS + t] = u4{vl[0], vl[1], vl[2], vl[3]};
            if constexpr (PIN_LO) asm volatile("" : "+a"(fh[tile * KS + t]), "+a"(fl[tile * KS + t]) :: "memory");
            else asm volatile("" : "+a"(fh[tile * KS + t]) :: "memory");
        }
}

struct NoSlot { template <typename I> __device__ __forceinline__ void operator()(I) const {} };
// One K-step: the A operand (vector forms, both fp16 pieces STACKED in its rows: chain_lane<PD, true>) against both pieces of R and the
// pieces of Q, two tiles, the four accumulators in rotation.  Round 4 read the two pieces as two operands and issued 12 MFMAs (8 in the QLITE
// instance); stacked it is 8 (6) -- v R_hi and v R_lo give hi hi' + lo hi' + hi lo' (+ lo lo', 2^-22 of the product: kept, it costs nothing) in registers
// 0 + 2 (own form) and 1 + 3 (partner) of the accumulator -- and one operand read instead of two.
template <int W, bool FIRST, bool QLITE = false, bool PIPE = false, typename Slot = NoSlot>
__device__ __forceinline__ void kstep_st(const u4& rh0, const u4& rl0, const u4& rh1, const u4& rl1, const u4& qh0, const u4& ql0, const u4& qh1,
                                         const u4& ql1, u4& v, f4& aR0, f4& aR1, f4& aQ0, f4& aQ1, Slot&& slot = NoSlot{}) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(W) : "memory");
    if constexpr (FIRST) { aR0 = f4{0.f, 0.f, 0.f, 0.f}; aR1 = aR0; aQ0 = aR0; aQ1 = aR0; }
    const h8 a = __builtin_bit_cast(h8, v);
#define C16_MMA(ACC, B) ACC = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, B), ACC, 0, 0, 0)
    if constexpr (PIPE) {               // the reverse scan: the slots' code first, then the MFMAs, interleaved 1 MFMA : 2 VALU by a sched_group_barrier pipeline instead of walls around every slot (-0.6 ms of 16.7 at C5 in round 4; no gain for the forward, whose slots hold LDS reads only)
        slot(ic<0>{}); slot(ic<1>{}); slot(ic<2>{}); slot(ic<3>{}); slot(ic<4>{}); slot(ic<5>{});
        C16_MMA(aR0, rh0); C16_MMA(aQ0, qh0); C16_MMA(aR1, rh1); C16_MMA(aQ1, qh1);
        C16_MMA(aR0, rl0); if constexpr (!QLITE) C16_MMA(aQ0, ql0); C16_MMA(aR1, rl1); if constexpr (!QLITE) C16_MMA(aQ1, ql1);
        mfma_valu_pipeline<QLITE ? 6 : 8>();
        __builtin_amdgcn_sched_barrier(0);
        return;
    }
    C16_MMA(aR0, rh0); C16_MMA(aQ0, qh0); __builtin_amdgcn_sched_barrier(0); slot(ic<0>{}); slot(ic<1>{}); __builtin_amdgcn_sched_barrier(0);
    C16_MMA(aR1, rh1); C16_MMA(aQ1, qh1); __builtin_amdgcn_sched_barrier(0); slot(ic<2>{}); slot(ic<3>{}); __builtin_amdgcn_sched_barrier(0);
    if constexpr (QLITE) {
        C16_MMA(aR0, rl0); C16_MMA(aR1, rl1); __builtin_amdgcn_sched_barrier(0); slot(ic<4>{}); slot(ic<5>{}); __builtin_amdgcn_sched_barrier(0);
    } else {
        C16_MMA(aR0, rl0); C16_MMA(aQ0, ql0); __builtin_amdgcn_sched_barrier(0); slot(ic<4>{}); __builtin_amdgcn_sched_barrier(0);
        C16_MMA(aR1, rl1); C16_MMA(aQ1, ql1); __builtin_amdgcn_sched_barrier(0); slot(ic<5>{}); __builtin_amdgcn_sched_barrier(0);
    }
#undef C16_MMA
}
// own + partner sums of a stacked accumulator: (hi-piece rows) + (lo-piece rows)
__device__ __forceinline__ f2 st_sum(const f4& c) { return f2{c[0], c[1]} + f2{c[2], c[3]}; }

// single 16-byte LDS reads (asm volatile statements keep their order; the data is valid after a matching counted wait)
template <int OFF, typename V>
__device__ __forceinline__ void rd128(unsigned addr, V& v) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=&v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ float gg_rsq_newton(float m) {      // as cmps_wide.hip::rsq_newton (the family's reverse scan and GEMM recompute it)
    const float r = __builtin_amdgcn_rsqf(m);
    return r * (1.5f - 0.5f * m * r * r);
}

// float offset of (row, component, clip) inside a vector of the wide family's stash (cmps_wide.hip: lane order of its chain kernels)
__device__ __forceinline__ int wide_pos(int row, int comp, int clip) {
    return 64 * (row >> 4) + 8 * (4 * ((row >> 3) & 1) + 2 * comp + clip) + (row & 7);
}

}  // namespace

// QLITE: the instance for |Q|_F <= 2^-19 (train.py's sigma = 1e-4 puts Q = -(dt sigma^2 / 2) R^dagger R near 1e-12): Q enters with its
// hi piece against the vector's hi piece only -- what is dropped is below 2^-11 |Q|_F |u| <= 2^-30 |u|, under the float32 rounding of
// u + Q u -- 8 instead of 12 MFMAs per K-step (round 5, stacked pieces: 6 instead of 8: Q's lo fragments are not multiplied).  Both instances are launched; each finds |Q|_F in its prologue and the one whose case
// it is not returns at once (the norm lives on the device: the host cannot choose).
template <int PD, bool QLITE>
__global__ __launch_bounds__(2 * PD, 1) void k_fwd_chain16(Dev P, const float* __restrict__ audio) {
    constexpr int PWV = PD / 32, KH = PD / 32, KS = PD / 16, VEC = Chain16Lds<PD>::VEC;
    __shared__ Chain16Lds<PD> L;
    __shared__ RhoStage<PD> RS;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int N = P.N, T = P.T, NC = (N + PCH - 1) / PCH;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;   // an odd batch repeats its last clip
    const bool two = b1 != b0;
    const float* xr0 = audio + (size_t)b0 * T;
    const float* xr1 = audio + (size_t)b1 * T;
    const float A = dev_A(P);
    if ((*P.qflag != 0u) != QLITE) return;        // the other instance's case (decided once per parameter set: k_qflag)

    // ---- operand scales: max |R|, max |Q|, Frobenius norms, max |s| over the pair's clips ----
    float sR, sQ, sV;
    {
        const int row0 = 32 * w + (lane & 31), c0 = (lane >> 5) * (PD / 2);
        float mR = 0.f, mQ = 0.f, fR = 0.f, fQ = 0.f, ms = 0.f;
        for (int c = 0; c < PD / 2; ++c) {
            const float2 r = P.R[(size_t)row0 * PD + c0 + c], qq = P.Q[(size_t)row0 * PD + c0 + c];
            mR = fmaxf(mR, fmaxf(fabsf(r.x), fabsf(r.y)));
            mQ = fmaxf(mQ, fmaxf(fabsf(qq.x), fabsf(qq.y)));
            fR += r.x * r.x + r.y * r.y;
            fQ += qq.x * qq.x + qq.y * qq.y;
        }
        for (int idx = threadIdx.x; idx < N; idx += 2 * PD) {
            ms = fmaxf(ms, fabsf((xr0[idx + 1] - xr0[idx]) / A));
            ms = fmaxf(ms, fabsf((xr1[idx + 1] - xr1[idx]) / A));
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mR = fmaxf(mR, __shfl_xor(mR, off, 64)); mQ = fmaxf(mQ, __shfl_xor(mQ, off, 64)); ms = fmaxf(ms, __shfl_xor(ms, off, 64));
            fR += __shfl_xor(fR, off, 64); fQ += __shfl_xor(fQ, off, 64);
        }
        if (lane == 0) { L.red[w][0] = mR; L.red[w][1] = mQ; L.red[w][2] = ms; L.nrm[0][0][w] = fR; L.nrm[0][1][w] = fQ; }
        __syncthreads();
        mR = mQ = ms = fR = fQ = 0.f;
#pragma unroll
        for (int ww = 0; ww < PWV; ++ww) {
            mR = fmaxf(mR, L.red[ww][0]); mQ = fmaxf(mQ, L.red[ww][1]); ms = fmaxf(ms, L.red[ww][2]);
            fR += L.nrm[0][0][ww]; fQ += L.nrm[0][1][ww];
        }
        __syncthreads();
        auto uni = [](float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
        sR = uni(gg::pow2_scale(mR, 15));
        sQ = uni(gg::pow2_scale(mQ, 15));
        sV = uni(gg::pow2_scale(1.01f * (1.0f + sqrtf(fQ) + ms * sqrtf(fR)), 13));
    }
    const float iRV = (1.0f / sR) * (1.0f / sV), iQV = (1.0f / sQ) * (1.0f / sV);    // exact: powers of two

    u4 FRh[PD / 8], FRl[PD / 8], FQh[PD / 8], FQl[QLITE ? 1 : PD / 8];
    {
        const int row0 = 32 * w + 2 * (lane & 15);
        const float2* Rrow = P.R + (size_t)row0 * PD;
        const float2* Qrow = P.Q + (size_t)row0 * PD;
        load_frags_f16<PD>(FRh, FRl, w, lane >> 4, sR, [&](int tile, int half, int c) { return half ? Rrow[tile * PD + c].y : Rrow[tile * PD + c].x; });
        if constexpr (QLITE) {
            u4 dump[PD / 8];
            load_frags_f16<PD, false>(FQh, dump, w, lane >> 4, sQ, [&](int tile, int half, int c) { return half ? Qrow[tile * PD + c].y : Qrow[tile * PD + c].x; });
            FQl[0] = u4{0u, 0u, 0u, 0u};
        } else {
            load_frags_f16<PD>(FQh, FQl, w, lane >> 4, sQ, [&](int tile, int half, int c) { return half ? Qrow[tile * PD + c].y : Qrow[tile * PD + c].x; });
        }
    }
    int lane_c = lane;
    asm volatile("" : "+v"(lane_c));
    const ChainLane<KH> g = chain_lane<PD, true>(w, lane_c, lds_addr_of(&L.vec[0][0][0]));
    const int q = g.q, ia = g.ia, ib = g.ib;
    const bool odd = g.odd;
    const float sg = odd ? 1.f : -1.f;
    const unsigned a_nrm = lds_addr_of(&L.nrm[0][0][0]);             // + 32 parity: (clip 0 row, clip 1 row)
    const unsigned a_rho = lds_addr_of(&RS.row[0][0][ia]);            // + 8 PD (32 buffer + row)
    const float2 pa = P.psi0[ia], pb = P.psi0[ib];
    f2 ua = odd ? f2{pa.y, pa.x} : f2{pa.x, pa.y}, ub = odd ? f2{pb.y, pb.x} : f2{pb.x, pb.y};
    float sv0 = 0.f, sv1 = 0.f;
    float nbuf0 = 0.f, nbuf1 = 0.f;                                   // wave 0: |y_k|^2 of the current 64-step chunk, lane <-> step
    float* stash = reinterpret_cast<float*>(P.stash) + (size_t)blockIdx.x * N * (8 * PD) + wide_pos(ia, odd ? 1 : 0, q);
    u4 o0, o1;                                                        // the wave's own K-steps: [K half] (both pieces stacked in the A rows)
    float n0p = 1.f, n1p = 1.f;                                       // |y|^2 of both clips as the last step's tail read them
    // |y_kn|^2 rows of the scalar stash (wave 0; lane <-> step, one row of 64 steps per chunk), a step late and off the chain's tail
    auto book = [&](int kn) {
        if (w != 0 || kn < 0 || kn > N - 1) return;
        if (lane_c == (kn & (PCH - 1))) { nbuf0 = n0p; nbuf1 = n1p; }
        if ((kn & (PCH - 1)) == PCH - 1 || kn == N - 1) {
            const int c = kn / PCH;
            if (c * PCH + lane_c < N) {
                P.scal[((size_t)b0 * NC + c) * 128 + lane_c] = nbuf0;
                if (two) P.scal[((size_t)b1 * NC + c) * 128 + lane_c] = nbuf1;
            }
        }
    };
    auto write_image = [&](int par, float xa, float xb) {
        unsigned hi, lo;
        split_f16x2(xa * sV, xb * sV, hi, lo);
        unsigned char* b0p = L.vec[par][0];
        unsigned char* b1p = L.vec[par][1];
        *reinterpret_cast<unsigned*>(b0p + g.wr1) = hi;
        *reinterpret_cast<unsigned*>(b0p + g.wr2) = hi ^ 0x80008000u;
        *reinterpret_cast<unsigned*>(b1p + g.wr2) = lo;               // piece 1: array kind k in the place of kind 3 - k (chain_lane)
        *reinterpret_cast<unsigned*>(b1p + g.wr1) = lo ^ 0x80008000u;
    };
    write_image(0, ua.x, ub.x);
    rd_own<0>(g.lo[0], g.hi[0], o0, o1);
    rho_stage<PD>(P, RS, 0, 0, 64 * w + lane_c);
    __syncthreads();

#if defined(CMPS_DIAG) && defined(C16_TIMING)              // diagnostic builds only: s_memtime stamps of a step's phases
    unsigned long long tS[6] = {0, 0, 0, 0, 0, 0}, tAcc[5] = {0, 0, 0, 0, 0}, tN = 0;
#define C16_STAMP(i) tS[i] = __builtin_readcyclecounter();
#define C16_STAMP_DEP(i, x) { asm volatile("" : "+v"(x)); tS[i] = __builtin_readcyclecounter(); }
#define C16_STAMPS_END() { for (int z = 0; z < 5; ++z) tAcc[z] += tS[z + 1] - tS[z]; ++tN; }
#else
#define C16_STAMP(i)
#define C16_STAMP_DEP(i, x)
#define C16_STAMPS_END()
#endif
#define QL_(i) FQl[QLITE ? 0 : (i)]
#define C16_STEP(PAR)                                                                                                      \
    {                                                                                                                      \
        constexpr int p = (PAR);                                                                                           \
        const int k = 2 * it + p;                                                                                          \
        C16_STAMP(0)                                                                                                       \
        if (p == 0 && (k & (PCH - 1)) == 0) {                          /* increments of the next 64 steps, one per lane */  \
            const int idx = k + lane_c;                                                                                    \
            const bool in0 = idx < T, in1 = idx + 1 < T;                                                                   \
            sv0 = ((in1 ? xr0[idx + 1] : 0.f) - (in0 ? xr0[idx] : 0.f)) / A;             /* model.py:263, 303 */            \
            sv1 = ((in1 ? xr1[idx + 1] : 0.f) - (in0 ? xr1[idx] : 0.f)) / A;                                               \
        }                                                                                                                  \
        if (p == 0 && (k & (RCH - 1)) == 0) rho_stage<PD>(P, RS, k / RCH + 1, (k / RCH + 1) & 1, 64 * w + lane_c);          \
        const unsigned ax0 = a_nrm + 32 * p;                                                                               \
        const unsigned ax2 = a_rho + 8 * PD * (((k / RCH) & 1) * RCH + (k & (RCH - 1)));                                   \
        f4 xn0, xn1, rh, cR0, cR1, cQ0, cQ1;                                                                               \
        u4 v[2 * KH - 2];                                                                                                  \
        /* the step's LDS reads (the other waves' K ranges, both pieces; three table rows) are issued two per K-step behind pairs of    \
           MFMAs, two K-steps ahead of their use: the four waves' 60 KB per step then pass the LDS (128 B / clk) spread over the step  \
           instead of in one burst behind the barrier (which took ~390 cycles with the matrix pipe idle, ~200 even when interleaved    \
           with the own K-steps' MFMAs).  K-step index 0, 1 = the own K ranges; index i + 2 = rest K-step i, operands v[2 i], v[2 i + 1]. */ \
        auto rd_slot = [&](auto kidx_c, auto sl_c) {                                                                       \
            constexpr int kidx = decltype(kidx_c)::value, sl = decltype(sl_c)::value, NR = 2 * KH - 2;                     \
            if constexpr (kidx < NR && sl == 0) {                                                                          \
                constexpr int half = kidx / (KH - 1), t = 1 + kidx % (KH - 1);                                             \
                rd128<p * 2 * VEC>(half ? g.hi[t] : g.lo[t], v[kidx]);                                                     \
            }                                                                                                              \
            if constexpr (kidx == NR && sl == 0) rd128<0>(ax0, xn0);                                                       \
            if constexpr (kidx == NR && sl == 2) rd128<16>(ax0, xn1);                                                      \
            if constexpr (kidx == NR && sl == 4) rd128<0>(ax2, rh);                                                        \
            if constexpr (kidx == 1 && sl == 5) book(k - 2);           /* wave 0: |y_{k-2}|^2 into the scalar stash's rows */ \
        };                                                                                                                 \
        kstep_st<15, true, QLITE>(FRh[0], FRl[0], FRh[KS], FRl[KS], FQh[0], QL_(0), FQh[KS], QL_(KS), o0, cR0, cR1, cQ0, cQ1,             \
                                  [&](auto sl) { rd_slot(ic<0>{}, sl); });                                                 \
        kstep_st<15, false, QLITE>(FRh[KH], FRl[KH], FRh[KS + KH], FRl[KS + KH], FQh[KH], QL_(KH), FQh[KS + KH], QL_(KS + KH), o1, cR0, cR1, cQ0, cQ1, \
                                   [&](auto sl) { rd_slot(ic<1>{}, sl); });                                                \
        C16_STAMP(1)                                                                                                       \
        gg::static_for<0, 2 * KH - 2>([&](auto ic_) {                                                                      \
            constexpr int I = decltype(ic_)::value, NR = 2 * KH - 2;                                                       \
            constexpr int T_ = I < KH - 1 ? 1 + I : KH + 1 + (I - (KH - 1));                                               \
            /* in flight behind this K-step's operand: the reads issued during the K-step before (one operand, or the three tables) */ \
            kstep_st<(I + 1 < NR ? 1 : 3), false, QLITE>(FRh[T_], FRl[T_], FRh[KS + T_], FRl[KS + T_], FQh[T_], QL_(T_), FQh[KS + T_],  \
                                                         QL_(KS + T_), v[I], cR0, cR1, cQ0, cQ1,                             \
                                                         [&](auto sl) { rd_slot(ic<I + 2>{}, sl); });                       \
        });                                                                                                                \
        C16_STAMP(2)                                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xn0), "+v"(xn1), "+v"(rh) :: "memory");                                  \
        C16_STAMP_DEP(3, cQ1[1])                                                                                           \
        const float n0 = sum4<PWV>(xn0), n1 = sum4<PWV>(xn1);                                                              \
        const float nq = q ? n1 : n0;                                                                                      \
        float inv = gg_rsq_newton(fmaxf(nq, 1e-12f));                                    /* model.py:332 */                \
        if (k == 0) inv = 1.f;                                                                                             \
        const float s0 = rdl(sv0, k & (PCH - 1)), s1 = rdl(sv1, k & (PCH - 1));                                            \
        const float sr = (q ? s1 : s0) * iRV;                                                                              \
        const f2 ya = inv * (ua + (st_sum(cQ0) * iQV + sr * st_sum(cR0)));                                                  \
        const f2 yb = inv * (ub + (st_sum(cQ1) * iQV + sr * st_sum(cR1)));                                                  \
        const f2 n2 = ya * ya + yb * yb;                                                                                   \
        float nn = n2.x + n2.y;                                                                                            \
        const f2 ta = f2{sg * rh.y, -(sg * rh.y)} * __builtin_shufflevector(ya, ya, 1, 0);                                  \
        const f2 tb = f2{sg * rh.w, -(sg * rh.w)} * __builtin_shufflevector(yb, yb, 1, 0);                                  \
        ua = rh.x * ya + ta;   ub = rh.z * yb + tb;                   /* ut_{k+1} = rho_k y_k (un-normalised), own and partner */ \
        write_image(p ^ 1, ua.x, ub.x);                                                                                    \
        rd_own<(p ^ 1) * 2 * VEC>(g.lo[0], g.hi[0], o0, o1);          /* (same wave, in order: no wait between store and read) */ \
        nn = row_sum16(nn);                                                                                                \
        if (lane_c == 0 || lane_c == 32) L.nrm[p ^ 1][q][w] = nn;                                                          \
        if (k < N) *reinterpret_cast<float2*>(stash + (size_t)k * (8 * PD)) = make_float2(ya.x, yb.x);                      \
        n0p = n0; n1p = n1;                                            /* |y_{k-1}|^2: booked behind the MFMAs of the next step */ \
        C16_STAMP_DEP(4, nn)                                                                                               \
        lds_barrier();                                                                                                     \
        C16_STAMP(5)                                                                                                       \
        C16_STAMPS_END()                                                                                                   \
    }
    for (int it = 0; it < (N + 2) / 2; ++it) {
        C16_STEP(0) C16_STEP(1)
    }
    book(2 * ((N + 2) / 2) - 2);                                      // (the last step's tail; beyond N - 1 when N is even: already booked)
#undef C16_STEP
#undef QL_
#if defined(CMPS_DIAG) && defined(C16_TIMING)
    if (blockIdx.x == 0 && lane == 0)
        printf("k_fwd_chain16 wave %d, cycles per step: reads issued + own K-steps issued %.1f | rest K-steps issued %.1f | tables + accumulators ready %.1f | "
               "tail -> barrier entry %.1f | barrier wait %.1f\n", w, (double)tAcc[0] / tN, (double)tAcc[1] / tN, (double)tAcc[2] / tN, (double)tAcc[3] / tN,
               (double)tAcc[4] / tN);
#endif
}

#undef C16_STAMP
#undef C16_STAMP_DEP
#undef C16_STAMPS_END

hipError_t launch_fwd_chain16(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    if (P.DP == 128) {
        hipLaunchKernelGGL((k_fwd_chain16<128, true>), dim3(nb), dim3(256), 0, s, P, audio);
        hipLaunchKernelGGL((k_fwd_chain16<128, false>), dim3(nb), dim3(256), 0, s, P, audio);
    } else if (P.DP == 96) {
        hipLaunchKernelGGL((k_fwd_chain16<96, true>), dim3(nb), dim3(192), 0, s, P, audio);
        hipLaunchKernelGGL((k_fwd_chain16<96, false>), dim3(nb), dim3(192), 0, s, P, audio);
    } else if (P.DP == 64) {
        hipLaunchKernelGGL((k_fwd_chain16<64, true>), dim3(nb), dim3(128), 0, s, P, audio);
        hipLaunchKernelGGL((k_fwd_chain16<64, false>), dim3(nb), dim3(128), 0, s, P, audio);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace cmps
namespace cmps {

// ------------------------------------------------------------------------------------------------
// reverse scan: the cotangent recursion; the rank-1 gradient contractions over (clip, step) are a GEMM (k_grad_gemm, cmps_grad_gemm.h) that builds
// its five bf16 operands itself from the float32 rows: this kernel only leaves ybar_k behind (Dev::gops: [pair][step][clip][re | im][D]
// float32, one 8-byte store per lane and step, 128 contiguous bytes per (clip, component)).  Round 2 wrote the five operands here,
// GEMM-ready in bf16 (ten 16-byte pieces per lane and eight steps: ~3 of the scan's 12.7 ms and 39 GB of traffic per step).
//   yhat = y_k inv_k;  yhb = conj(rho_k) g;  ybar = (yhb - yhat rad_{k+1}) inv_k + te_k H y_k
//   g    = ybar + Q ybar + s_k R^dagger ybar            (4 D / 16 16x16x32 MFMAs, bf16 operands)
//   fbar += dt_k Im(g conj(rho_k yhat));   Abar += zbar_k (-e_k x_k / A^2) + Re(d^dagger u_k) (-x_k / A^2)
// rad_{k+1} = Re(u_{k+1}^dagger g_{k+1}) = 2 ebar_{k+1} e_{k+1} (Euler's theorem: everything downstream of u_{k+1} is
// scale invariant except the loss term of step k+1, which is homogeneous of degree 2), 0 behind the last step.
// Only  g -> conj(rho) g -> ybar -> bf16 -> LDS -> barrier -> mat-vec -> g  is on the chain: the part of ybar that does
// not depend on g (c3 = te H y - ok yhat rad inv), u_k, the scalars and rho row of the next step are computed / fetched in
// the shadow of the LDS reads and MFMAs of the step before (the pieces of matvec2).
// ------------------------------------------------------------------------------------------------
namespace {

// per-step scalars of one clip: (s, inv, ok, te | rad, xa = -x/A^2, dt, -)
template <int W>
struct StepTab {
    __attribute__((aligned(16))) f4 row[W][2][PCH][2][2];              // [wave][chunk parity][step][clip][half]
};

constexpr int GB = 8;          // steps per block of the unrolled sweep
#ifndef PAIR_BWD_TL
#define PAIR_BWD_TL 1
#endif
constexpr bool PAIR_BWD_TABLES_LAST = PAIR_BWD_TL != 0;     // the scalar rows of step k - 2 behind the operand reads (A/B: -DPAIR_BWD_TL=0)
#if defined(CMPS_DIAG) && defined(PABL_NO_PIECES)     // diagnostic builds only (scripts/ablate.py): the reverse scan without its off-chain work
constexpr bool PAIR_NO_PIECES = true;
#else
constexpr bool PAIR_NO_PIECES = false;
#endif

}  // namespace

template <int PD>
__global__ __launch_bounds__(2 * PD, 1) void k_bwd_pair(Dev P, const float* __restrict__ audio) {
    constexpr int PWV = PD / 32;
    __shared__ PairLds<PD> L;
    __shared__ StepTab<PWV> TB;
    __shared__ RhoStage<PD> RS;
    __shared__ float redA[PWV];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    const int N = P.N, T = P.T, NC = (N + PCH - 1) / PCH;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;
    const bool two = b1 != b0;

    u4 FQ[PD / 8], FD[PD / 8];                                        // Q (Hermitian) and R^dagger
    {
        const int row0 = 32 * w + 2 * (lane0 & 15);                    // rows ia (tile 0) and ia + 1 (tile 1)
        const float2* Qrow = P.Q + (size_t)row0 * PD;
        const float2* RTrow = P.RT + (size_t)row0 * PD;                // R^dagger[i][j] = conj(R[j][i]) = conj(RT[i][j])
        load_frags<PD, true>(FQ, w, lane0 >> 4, [&](int tile, int half, int c) { return half ? Qrow[tile * PD + c].y : Qrow[tile * PD + c].x; });
        load_frags<PD, true>(FD, w, lane0 >> 4, [&](int tile, int half, int c) { return half ? -RTrow[tile * PD + c].y : RTrow[tile * PD + c].x; });
    }
    // (everything below is derived from a laundered copy of the lane number: see k_fwd_pair)
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    constexpr int KH = PD / 32, VEC = PairLds<PD>::VEC_BYTES;
    const ChainLane<KH> g = chain_lane<PD>(w, lane, lds_addr_of(&L.vec[0][0]));
    const int q = g.q, ia = g.ia, ib = g.ib;
    const bool odd = g.odd;
    const float wq = (q == 0 || two) ? 1.f : 0.f;                      // weight of this lane's clip (0: the repeated clip)
    const float* xr0 = audio + (size_t)b0 * T;
    const float* xr1 = audio + (size_t)b1 * T;
    const float* sc0 = P.scal + ((size_t)b0 * NC) * 128;
    const float* sc1 = P.scal + ((size_t)b1 * NC) * 128;
    const float* stf = reinterpret_cast<const float*>(P.stash);
    const int NBLK = (N + GB - 1) / GB;
    // ybar_k of this lane: rows ia, ia + 1 (8 contiguous bytes) of component (c & 1), clip q
    float* ybar_base = reinterpret_cast<float*>(P.gops) + ((size_t)blockIdx.x * N * 4 + (q * 2 + (odd ? 1 : 0))) * PD + ia;
    const float A = dev_A(P);
    const float sgn = odd ? 1.f : -1.f;                                // (rho x)_own = rho_re x_own + sgn rho_im x_partner

    // every wave builds its own copy of the chunk's scalar rows (no cross-wave hand-over): lane <-> step
    float accA = 0.f;
    auto chunk_rows = [&](int cj) {
        const int idx = cj * PCH + lane;
        const bool in = idx < N;
        const float dtv = in ? P.dtk[idx] : 0.f;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const float* xr = qq ? xr1 : xr0;
            const float* sc = qq ? sc1 : sc0;
            const float x0 = idx < T ? xr[idx] : 0.f, x1 = idx + 1 < T ? xr[idx + 1] : 0.f;
            const float inc = x1 - x0;
            const float nv = in ? sc[(size_t)cj * 128 + lane] : 1.f;
            const float ev = in ? sc[(size_t)cj * 128 + 64 + lane] : 0.f;
            const float ex = ev * inc;                                 // model.py:294 operation order
            const float z = ex / A;
            const float zbar = -1.0f / (1.0f + z);
            const float te = 2.0f * (zbar * inc / A);
            f4 r0, r1;
            r0.x = inc / A;
            r0.y = 1.0f / sqrtf(fmaxf(nv, 1e-12f));
            r0.z = nv > 1e-12f ? 1.f : 0.f;
            r0.w = te;
            r1.x = te * ev;
            r1.y = -inc / (A * A);
            r1.z = dtv;
            r1.w = 0.f;
            TB.row[w][cj & 1][lane][qq][0] = r0;
            TB.row[w][cj & 1][lane][qq][1] = r1;
            if (in && w == 0 && (qq == 0 || two)) accA += zbar * (-ex / (A * A));
        }
    };

    chunk_rows((N - 1) / PCH);
    if ((N - 1) / PCH > 0 && ((N - 1) & (PCH - 1)) == 0) chunk_rows((N - 1) / PCH - 1);   // step N - 2 lives in the chunk below
    // rho: 32-step chunks staged in LDS, descending; chunk j lives in buffer j & 1 and is loaded when the sweep enters
    // chunk j + 1 (rows k and k - 1 of a step can straddle two chunks, so two are always resident)
    {
        const int cl = (N - 1) / RCH;
        rho_stage<PD>(P, RS, cl, cl & 1, 64 * w + lane);
        if (cl > 0) rho_stage<PD>(P, RS, cl - 1, (cl - 1) & 1, 64 * w + lane);
    }
    __syncthreads();

    // ---- state of the step about to run (k): everything the chain needs before the barrier is in registers ----
    float ga = 0.f, gb = 0.f, pga = 0.f, pgb = 0.f;     // g: cotangent of u_{k+1}, own component and a copy of the partner's (MFMA register 1)
    float una = 0.f, unb = 0.f, puna = 0.f, punb = 0.f; // u_{k+1} = rho_k yhat_k, own and partner component (carried: it is u_k of the step before)
    float facca = 0.f, faccb = 0.f, accS = 0.f;
    float sda = 0.f, sdb = 0.f, ssy = 0.f;              // (R^dagger ybar) rows and the -x / A^2 factor of the step before: its Abar term is added a step late
    u4 vlo, vhi;                                        // the wave's own K-steps of ybar_k, read back before the barrier
    float4 rh, rhp;                             // rho_k and rho_{k-1}, rows ia | ib
    f4 S0, S1, SP0, SP1;                        // scalar rows of steps k and k - 1 (fetched two steps ahead, behind the MFMAs)
    float c3a, c3b;                             // te_k (H y_k) - ok_k yhat_k rad_{k+1} inv_k
    auto rho_rows = [&](int k) { return *reinterpret_cast<const float4*>(&RS.row[(k / RCH) & 1][k & (RCH - 1)][ia]); };
    auto tab_row = [&](int k, int half) { return TB.row[w][(k / PCH) & 1][k & (PCH - 1)][q][half]; };
    // unconditional (clamped) loads: a select on the loaded value would force the wait right behind the load
    // row k of the stash for this lane: (y_k a, y_k b) and ((H y_k) a, (H y_k) b), rows ia | ia + 1 of component (c & 1), clip q
    auto row_at = [&](int k) {
        const int kc = k > 0 ? k : 0;
        const float2 y = *reinterpret_cast<const float2*>(stf + pair_stash_index<PD>(blockIdx.x, N, kc, 0, q, odd ? 1 : 0, ia));
        const float2 h = *reinterpret_cast<const float2*>(stf + pair_stash_index<PD>(blockIdx.x, N, kc, 1, q, odd ? 1 : 0, ia));
        return make_float4(y.x, y.y, h.x, h.y);
    };
    // ring of eight stash rows: slot (k & 7) holds row k = (y_k a, y_k b, (H y_k) a, (H y_k) b), fetched seven steps (~5 us)
    // before its first use: under load the stash stream's latency is several microseconds (with a ring of four the scan
    // spent 665 of its 1744 cycles per step in s_waitcnt)
    float4 ring0, ring1, ring2, ring3, ring4, ring5, ring6, ring7;
    {
        const int k0 = N - 1;
        ring0 = row_at(k0 - ((k0 - 0) & 7));
        ring1 = row_at(k0 - ((k0 - 1) & 7));
        ring2 = row_at(k0 - ((k0 - 2) & 7));
        ring3 = row_at(k0 - ((k0 - 3) & 7));
        ring4 = row_at(k0 - ((k0 - 4) & 7));
        ring5 = row_at(k0 - ((k0 - 5) & 7));
        ring6 = row_at(k0 - ((k0 - 6) & 7));
        ring7 = row_at(k0 - ((k0 - 7) & 7));
        const int k1 = k0 > 0 ? k0 - 1 : 0;
        rh = rho_rows(k0);
        S0 = tab_row(k0, 0);
        S1 = tab_row(k0, 1);
        rhp = rho_rows(k1);
        SP0 = tab_row(k1, 0);
        SP1 = tab_row(k1, 1);
        const float4 cur = row_at(k0);              // (a second fetch of that row: selecting the slot at run time put the ring into scratch memory)
        c3a = S0.w * cur.z;                     // rad_N = 0
        c3b = S0.w * cur.w;
    }
    const float2 psa = P.psi0[ia], psb = P.psi0[ib];
    const float ps0a = odd ? psa.y : psa.x, ps0b = odd ? psb.y : psb.x;        // u_0 = psi_0: own component ..
    const float pps0a = odd ? psa.x : psa.y, pps0b = odd ? psb.x : psb.y;      // .. and the partner's
    const unsigned a_tab = lds_addr_of(&TB.row[w][0][0][q][0]);                 // + 64 (64 chunk parity + step in chunk)
#if defined(CMPS_DIAG) && defined(PABL_TIMING)        // diagnostic builds only: where a step's cycles go.  Fourteen s_memtime stamps per step into
    // SGPR pairs, consumed behind one wait at the end of the step.  (SMEM answers count in lgkmcnt and return out of order: the counted
    // LDS waits of such a build can let an operand through early -- its results are not to be used, its timings are representative.)
    constexpr int NST = 14;
    unsigned long long st[NST], sacc[NST] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stprev = 0;
    int stn = 0;
#define PAIR_STAMP(i) asm volatile("s_memtime %0" : "=s"(st[i]))
#define PAIR_STAMPS_END()                                                                                                          \
    {                                                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(st[0]), "+s"(st[1]), "+s"(st[2]), "+s"(st[3]), "+s"(st[4]), "+s"(st[5]), "+s"(st[6]), \
                     "+s"(st[7]), "+s"(st[8]), "+s"(st[9]), "+s"(st[10]), "+s"(st[11]), "+s"(st[12]), "+s"(st[13]) :: "memory");      \
        if (stprev) { sacc[0] += st[0] - stprev; for (int i_ = 1; i_ < NST; ++i_) sacc[i_] += st[i_] - st[i_ - 1]; ++stn; }          \
        stprev = st[NST - 1];                                                                                                      \
    }
#else
#define PAIR_STAMP(i)
#define PAIR_STAMPS_END()
#endif
    // one step; J = k & 7 (static: selects ring slots and the image parity).  COND: `true` in full blocks.
    // Chain: g -> conj(rho) g -> ybar -> bf16 image (own K ranges read back) -> [barrier] -> mat-vecs -> g.  Everything else is placed
    // where the chain wave would otherwise idle or where the matrix pipe covers it:
    //   * behind the image stores, while they complete (the barrier's lgkmcnt(0)): the ybar store, the partner copy of ybar, the
    //     frequency gradient and the Abar term of the step before (~20 instructions against ~100 cycles of LDS store latency);
    //   * behind the K-steps (about two instructions per MFMA are free for a lone wave): yhat_{k-1} and its partner (with ybar's the
    //     only two lane exchanges of a step), u_k (= the u_{k+1} of the next step: carried, not recomputed), the g-independent part
    //     of ybar_{k-1}, the stash ring's next row, the rho row of step k - 2 (its scalar rows come with the operand reads).
#define PAIR_BWD_STEP(J, CUR, PRV, COND)                                                                                            \
    if (COND) {                                                                                                                    \
        constexpr int p = (J) & 1;                                                                                                 \
        const int k = 8 * blk + (J);                                                                                               \
        const int km2 = k > 1 ? k - 2 : 0;                                                                                         \
        if ((k & (PCH - 1)) == 1 && k > 1) chunk_rows(k / PCH - 1);                 /* step k - 2: the chunk below */               \
        if ((k & (RCH - 1)) == RCH - 1 && k != N - 1 && k >= RCH)                   /* entering rho chunk k / RCH: fetch the one below */ \
            rho_stage<PD>(P, RS, k / RCH - 1, (k / RCH - 1) & 1, 64 * w + lane);                                                   \
        float4 nrh;                                                                                                                \
        f4 nS0, nS1;                                                                                                               \
        PAIR_STAMP(0);                                                                                                             \
        /* ---- the chain ---- */                                                                                                  \
        const float hba = rh.x * ga - sgn * rh.y * pga;              /* conj(rho_k) g */                                           \
        const float hbb = rh.z * gb - sgn * rh.w * pgb;                                                                            \
        float yba = fmaf(hba, S0.y, c3a), ybb = fmaf(hbb, S0.y, c3b);                                                              \
        PAIR_PIN2(yba, ybb);                                                                                                       \
        PAIR_STAMP(1);                                                                                                             \
        write_vec(L.vec[p], g, yba, ybb);                                                                                          \
        rd_own<p * VEC>(g.lo[0], g.hi[0], vlo, vhi);                                                                               \
        PAIR_STAMP(2);                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        /* ---- behind the stores (about 70 cycles of store latency are free here; more would delay the barrier) ---- */            \
        float pyba, pybb;                                                                                                          \
        const float una_ = una, unb_ = unb;                          /* u_{k+1}: the pieces below overwrite una .. only at the step's end */ \
        if constexpr (!PAIR_NO_PIECES) {                                                                                           \
            *reinterpret_cast<float2*>(ybar_base + (size_t)k * (4 * PD)) = make_float2(yba, ybb);    /* for the gradient GEMM */   \
            facca += S1.z * (pga * una - ga * puna);                  /* the frequency gradient (meaningful in the Re lanes) */    \
            faccb += S1.z * (pgb * unb - gb * punb);                                                                               \
        } else { pyba = yba; pybb = ybb; }                                                                                         \
        PAIR_PIN2(facca, faccb);                                                                                                   \
        PAIR_STAMP(3);                                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        lds_barrier();                                                                                                             \
        PAIR_STAMP(4);                                                                                                             \
        f4 cQ0, cQ1, cD0, cD1;                                                                                                     \
        float uka, ukb, puka, pukb, ypa, ypb, pypa, pypb;                                                                          \
        const unsigned ax0 = a_tab + 64 * (((km2 / PCH) & 1) * PCH + (km2 & (PCH - 1)));                                           \
        matvec2<PD, p * VEC, PAIR_BWD_TABLES_LAST>(FQ, FD, g, ax0, ax0 + 16, vlo, vhi, nS0, nS1, cQ0, cQ1, cD0, cD1, [&](auto pc) {                      \
            constexpr int PI = decltype(pc)::value;                                                                                \
            const float invp = SP0.y;                                                                                              \
            if constexpr (PI < 7) { PAIR_STAMP(5 + PI); }                                                                          \
            if constexpr (PAIR_NO_PIECES) {          /* diagnostic builds only */                                                  \
                if constexpr (PI == 0) { ypa = PRV.x; ypb = PRV.y; pypa = ypa; pypb = ypb; uka = ypa; ukb = ypb; puka = ypa; pukb = ypb; nrh = rhp; } \
            } else if constexpr (PI == 0) {          /* yhat_{k-1}, own and partner component (behind the eight own-range MFMAs) */ \
                ypa = PRV.x * invp; ypb = PRV.y * invp;                                                                            \
                pypa = partner16(ypa, odd); pypb = partner16(ypb, odd);                                                            \
                PAIR_PIN4(ypa, ypb, pypa, pypb);                                                                                   \
            } else if constexpr (PI == 1) {          /* u_k = rho_{k-1} yhat_{k-1}, both components: row ia .. */                  \
                const float ria = sgn * rhp.y;                                                                                     \
                uka = rhp.x * ypa + ria * pypa;   puka = rhp.x * pypa - ria * ypa;                                                 \
                PAIR_PIN2(uka, puka);                                                                                              \
            } else if constexpr (PI == 2) {          /* .. and row ib */                                                           \
                const float rib = sgn * rhp.w;                                                                                     \
                ukb = rhp.z * ypb + rib * pypb;   pukb = rhp.z * pypb - rib * ypb;                                                 \
                PAIR_PIN2(ukb, pukb);                                                                                              \
            } else if constexpr (PI == 3) {          /* the g-independent part of ybar_{k-1} */                                    \
                const float radk = S1.x * SP0.z * invp;                /* rad_k ok_{k-1} inv_{k-1} */                              \
                c3a = fmaf(SP0.w, PRV.z, -(ypa * radk));                                                                           \
                c3b = fmaf(SP0.w, PRV.w, -(ypb * radk));                                                                           \
                PAIR_PIN2(c3a, c3b);                                                                                               \
            } else if constexpr (PI == 4) {          /* the partner's ybar_k (row ia) and Re(d^dagger u) of the step before (its u_k is this step's u_{k+1}) */ \
                pyba = partner16(yba, odd);                                                                                        \
                accS += (sda * una_ + sdb * unb_) * ssy;                                                                           \
                PAIR_PIN2(pyba, accS);                                                                                             \
            } else if constexpr (PI == 5) {                                                                                        \
                pybb = partner16(ybb, odd);                                                                                        \
                PAIR_PIN1(pybb);                                                                                                   \
                CUR = row_at(k - 8);                                  /* this slot's next row (row k is dead from here on) */      \
            } else if constexpr (PI == 6) {                                                                                        \
                nrh = rho_rows(km2);                                  /* rho row of step k - 2 */                                  \
            }                                                                                                                      \
        });                                                                                                                        \
        PAIR_STAMP(12);                                                                                                            \
        if (k == 0) { uka = ps0a; ukb = ps0b; puka = pps0a; pukb = pps0b; }         /* u_0 = psi_0 */                              \
        {   /* g = ybar + Q ybar + s R^dagger ybar, (own, partner) pairs: packed float32 (the matrix pipe is idle here) */          \
            const f2 Ga = (f2{yba, pyba} + f2{cQ0[0], cQ0[1]}) + S0.x * f2{cD0[0], cD0[1]};                                        \
            const f2 Gb = (f2{ybb, pybb} + f2{cQ1[0], cQ1[1]}) + S0.x * f2{cD1[0], cD1[1]};                                        \
            ga = Ga.x; pga = Ga.y; gb = Gb.x; pgb = Gb.y;                                                                          \
        }                                                                                                                          \
        sda = cD0[0]; sdb = cD1[0]; ssy = S1.y;                                                                                    \
        una = uka; unb = ukb; puna = puka; punb = pukb;                                                                            \
        rh = rhp; S0 = SP0; S1 = SP1;                                                                                              \
        rhp = nrh; SP0 = nS0; SP1 = nS1;                                                                                           \
        PAIR_PIN4(ga, gb, pga, pgb);                                                                                               \
        PAIR_STAMP(13);                                                                                                            \
        PAIR_STAMPS_END();                                                                                                         \
    }

    int blk = NBLK - 1;
    if (N & (GB - 1)) {                                                // the partly filled top block
        PAIR_BWD_STEP(7, ring7, ring6, 8 * blk + 7 < N)
        PAIR_BWD_STEP(6, ring6, ring5, 8 * blk + 6 < N)
        PAIR_BWD_STEP(5, ring5, ring4, 8 * blk + 5 < N)
        PAIR_BWD_STEP(4, ring4, ring3, 8 * blk + 4 < N)
        PAIR_BWD_STEP(3, ring3, ring2, 8 * blk + 3 < N)
        PAIR_BWD_STEP(2, ring2, ring1, 8 * blk + 2 < N)
        PAIR_BWD_STEP(1, ring1, ring0, 8 * blk + 1 < N)
        PAIR_BWD_STEP(0, ring0, ring7, true)
        --blk;
    }
    for (; blk >= 0; --blk) {
        PAIR_BWD_STEP(7, ring7, ring6, true)
        PAIR_BWD_STEP(6, ring6, ring5, true)
        PAIR_BWD_STEP(5, ring5, ring4, true)
        PAIR_BWD_STEP(4, ring4, ring3, true)
        PAIR_BWD_STEP(3, ring3, ring2, true)
        PAIR_BWD_STEP(2, ring2, ring1, true)
        PAIR_BWD_STEP(1, ring1, ring0, true)
        PAIR_BWD_STEP(0, ring0, ring7, true)
    }
#undef PAIR_BWD_STEP
#if defined(CMPS_DIAG) && defined(PABL_TIMING)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        printf("reverse scan, cycles per step by phase (0 loop overhead | 1 chain VALU | 2 image stores + own reads issued | 3 behind-the-stores work | "
               "4 wait + barrier | 5 operand reads issued + own K-steps issued | 6..11 pieces / K-steps 1.. | 12 last K-steps | 13 tail):\n  cycles per step:");
        double tot = 0;
        for (int i_ = 0; i_ < NST; ++i_) { printf(" %.0f", (double)sacc[i_] / stn); tot += (double)sacc[i_] / stn; }
        printf("   total %.0f\n", tot);
    }
#endif
#undef PAIR_STAMP
#undef PAIR_STAMPS_END
    accS += (sda * una + sdb * unb) * ssy;                             // the Abar term of step 0
    // ---- the pair's slab: f | psi0bar_re | psi0bar_im | A (the R / Q sections are written by k_grad_gemm) ----
    float* slab = P.slabs + (size_t)blockIdx.x * P.slab_floats;
    const int DD = PD * PD;
    {
        const float fa = facca * wq, fb = faccb * wq;
        const float fta = both_clips(fa), ftb = both_clips(fb);
        const float g0a = ga * wq, g0b = gb * wq;
        const float gta = both_clips(g0a), gtb = both_clips(g0b);
        if (g.f == 0) {                                                    // Re lanes of clip 0
            slab[4 * DD + ia] = fta;
            slab[4 * DD + ib] = ftb;
            slab[4 * DD + PD + ia] = gta;
            slab[4 * DD + PD + ib] = gtb;
        } else if (g.f == 1) {                                             // Im lanes of clip 0
            slab[4 * DD + 2 * PD + ia] = gta;
            slab[4 * DD + 2 * PD + ib] = gtb;
        }
    }
    // Abar: sum of accS over all lanes (both clips, weighted) + sum of accA over wave 0's lanes
    {
        float t = accS * wq + (w == 0 ? accA : 0.f);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        if (lane == 0) redA[w] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
#pragma unroll
            for (int i = 0; i < PWV; ++i) tot += redA[i];
            slab[4 * DD + 3 * PD] = tot;
            slab[4 * DD + 3 * PD + 1] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// float32-faithful reverse chain on the matrix cores (round 4): the wide family's reverse scan when CMPS_OPT_WIDE_CHAIN = MFMA.
// k_bwd_pair's lane layout and step structure with k_bwd_wide's arithmetic (its per-step scalars, rows, slab and Abar conventions)
// and k_fwd_chain16's operands: two fp16 pieces per value, three products (Q: one in the QLITE instance); round 5: 8 (6) MFMAs per K-step (kstep_st).
// What is new here is the scale of the broadcast vector: ybar_k has no a-priori size (in the forward |ut| <= 1 + |M|), and an fp16
// overflow would be silent.  Every step derives a power of two per clip from a GUARANTEED bound, identical in all lanes and known
// before the image is written:
//   Y_{j} <= a_j Y_{j+1} + c_j,   a_j = inv_j sqrt2 (1 + |Q|_inf + |s_{j+1}| |R^dagger|_inf),   c_j = |te_j| |H|_inf |y_j| + |rad_{j+1}| ok_j inv_j
// applied twice: Y_{k-2} <= a_{k-2} (a_{k-1} Y_k + c_{k-1}) + c_{k-2}  (Y_j = max |ybar_j|; row-sum norms from the prologue; Y_k is the
// MEASURED maximum of two steps before: every wave leaves its own in LDS behind the MFMAs of step k, the barrier of step k - 1
// publishes it, and it is used for the image written at the top of step k - 2 -- nothing of this sits on the chain; the a's and c's
// are the steps' table scalars).  The bound is loose by the usual norm factors (2^4 .. 2^6), which
// costs nothing: pieces are exact down to 2^-18 of the scaled bound.
// ------------------------------------------------------------------------------------------------
template <int PD, bool QLITE>
__global__ __launch_bounds__(2 * PD, 1) void k_bwd_chain16(Dev P, const float* __restrict__ audio) {
    constexpr int PWV = PD / 32, KH = PD / 32, KS = PD / 16, VEC = Chain16Lds<PD>::VEC, NR = 2 * KH - 2;
    __shared__ Chain16Lds<PD> L;
    __shared__ StepTab<PWV> TB;
    __shared__ RhoStage<PD> RS;
    __shared__ __attribute__((aligned(16))) unsigned ymx_tab[2][2][4][2];  // [step parity][clip][wave][component]: max |ybar| of the wave's lanes (float bits)
    __shared__ float redA[PWV];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane0 = threadIdx.x & 63;
    const int N = P.N, T = P.T, NC = (N + PCH - 1) / PCH;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;
    const bool two = b1 != b0;
    const float A = dev_A(P);
    if ((*P.qflag != 0u) != QLITE) return;        // as k_fwd_chain16

    // ---- matrix scales and norms: max entries (-> fp16 scales), Frobenius norm of Q, row sums of |Q|, |R^dagger|, |H| ----
    float sQ, sD, Qinf, Dinf, Hinf;
    {
        const int row0 = 32 * w + (lane0 & 31), c0 = (lane0 >> 5) * (PD / 2);
        float mQ = 0.f, mD = 0.f, fQ = 0.f, rQ = 0.f, rD = 0.f, rH = 0.f;
        for (int c = 0; c < PD / 2; ++c) {
            const float2 qq = P.Q[(size_t)row0 * PD + c0 + c], rt = P.RT[(size_t)row0 * PD + c0 + c], r = P.R[(size_t)row0 * PD + c0 + c];
            mQ = fmaxf(mQ, fmaxf(fabsf(qq.x), fabsf(qq.y)));
            mD = fmaxf(mD, fmaxf(fabsf(rt.x), fabsf(rt.y)));
            fQ += qq.x * qq.x + qq.y * qq.y;
            rQ += sqrtf(qq.x * qq.x + qq.y * qq.y);
            rD += sqrtf(rt.x * rt.x + rt.y * rt.y);
            rH += sqrtf((r.x + rt.x) * (r.x + rt.x) + (r.y - rt.y) * (r.y - rt.y));      // H = R + R^dagger
        }
        rQ += __shfl_xor(rQ, 32, 64); rD += __shfl_xor(rD, 32, 64); rH += __shfl_xor(rH, 32, 64);     // the two halves of a row
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mQ = fmaxf(mQ, __shfl_xor(mQ, off, 64)); mD = fmaxf(mD, __shfl_xor(mD, off, 64));
            rQ = fmaxf(rQ, __shfl_xor(rQ, off, 64)); rD = fmaxf(rD, __shfl_xor(rD, off, 64)); rH = fmaxf(rH, __shfl_xor(rH, off, 64));
            fQ += __shfl_xor(fQ, off, 64);
        }
        if (lane0 == 0) { L.red[w][0] = mQ; L.red[w][1] = mD; L.red[w][2] = fQ; L.nrm[0][0][w] = rQ; L.nrm[0][1][w] = rD; L.nrm[1][0][w] = rH; }
        __syncthreads();
        mQ = mD = fQ = rQ = rD = rH = 0.f;
#pragma unroll
        for (int ww = 0; ww < PWV; ++ww) {
            mQ = fmaxf(mQ, L.red[ww][0]); mD = fmaxf(mD, L.red[ww][1]); fQ += L.red[ww][2];
            rQ = fmaxf(rQ, L.nrm[0][0][ww]); rD = fmaxf(rD, L.nrm[0][1][ww]); rH = fmaxf(rH, L.nrm[1][0][ww]);
        }
        __syncthreads();
        auto uni = [](float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
        sQ = uni(gg::pow2_scale(mQ, 15));
        sD = uni(gg::pow2_scale(mD, 15));
        Qinf = uni(1.001f * rQ); Dinf = uni(1.001f * rD); Hinf = uni(1.001f * rH);
    }
    const float iQ = 1.0f / sQ, iD = 1.0f / sD;

    u4 FQh[PD / 8], FQl[QLITE ? 1 : PD / 8], FDh[PD / 8], FDl[PD / 8];  // Q (Hermitian) and R^dagger
    {
        const int row0 = 32 * w + 2 * (lane0 & 15);
        const float2* Qrow = P.Q + (size_t)row0 * PD;
        const float2* RTrow = P.RT + (size_t)row0 * PD;                // R^dagger[i][j] = conj(RT[i][j])
        load_frags_f16<PD>(FDh, FDl, w, lane0 >> 4, sD, [&](int tile, int half, int c) { return half ? -RTrow[tile * PD + c].y : RTrow[tile * PD + c].x; });
        if constexpr (QLITE) {
            u4 dump[PD / 8];
            load_frags_f16<PD, false>(FQh, dump, w, lane0 >> 4, sQ, [&](int tile, int half, int c) { return half ? Qrow[tile * PD + c].y : Qrow[tile * PD + c].x; });
            FQl[0] = u4{0u, 0u, 0u, 0u};
        } else {
            load_frags_f16<PD>(FQh, FQl, w, lane0 >> 4, sQ, [&](int tile, int half, int c) { return half ? Qrow[tile * PD + c].y : Qrow[tile * PD + c].x; });
        }
    }
    int lane = lane0;
    asm volatile("" : "+v"(lane));
    const ChainLane<KH> g = chain_lane<PD, true>(w, lane, lds_addr_of(&L.vec[0][0][0]));
    const int q = g.q, ia = g.ia, ib = g.ib;
    const bool odd = g.odd;
    const float wq = (q == 0 || two) ? 1.f : 0.f;
    const float* xr0 = audio + (size_t)b0 * T;
    const float* xr1 = audio + (size_t)b1 * T;
    const float* sc0 = P.scal + ((size_t)b0 * NC) * 128;
    const float* sc1 = P.scal + ((size_t)b1 * NC) * 128;
    const int wpos = wide_pos(ia, odd ? 1 : 0, q);
    // rows through buffer instructions: descriptor + SGPR row offset + this lane's loop-invariant offset (no address arithmetic on the VALU)
    const auto rs_st = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(P.stash) + (size_t)blockIdx.x * N * (8 * PD), 0, N * (8 * PD * 4), 0x00020000);
    const auto rs_yb = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(P.gops) + (size_t)blockIdx.x * N * (4 * PD), 0, N * (4 * PD * 4), 0x00020000);
    const int voff = wpos * 4;
    const float sgn = odd ? 1.f : -1.f;

    // per-step scalars exactly as cmps_wide.hip::step_scalars forms them (the gradient GEMM recomputes them the same way)
    float accA = 0.f;
    const float gr0 = 1.4143f * (1.0f + Qinf), gr1 = 1.4143f * Dinf;
    float s_ab0 = 0.f, s_ab1 = 0.f, rad_ab0 = 0.f, rad_ab1 = 0.f;      // s, rad of the lowest step of the chunk built before (the one above; no step N: 0)
    auto chunk_rows = [&](int cj) {
        const int idx = cj * PCH + lane;
        const bool in = idx < N;
        const float dtv = in ? P.dtk[idx] : 0.f;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const float* xr = qq ? xr1 : xr0;
            const float* sc = qq ? sc1 : sc0;
            const float x0 = idx < T ? xr[idx] : 0.f, x1 = idx + 1 < T ? xr[idx + 1] : 0.f;
            const float inc = x1 - x0;
            const float nv = in ? sc[(size_t)cj * 128 + lane] : 1.f;
            const float ev = in ? sc[(size_t)cj * 128 + 64 + lane] : 0.f;
            const float ex = ev * inc;
            const float z = ex / A;
            const float zbar = -1.0f / (1.0f + z);
            f4 r0, r1;
            r0.x = inc / A;
            r0.y = gg_rsq_newton(fmaxf(nv, 1e-12f));
            r0.z = nv > 1e-12f ? 1.f : 0.f;
            r0.w = 2.0f * (zbar * inc / A);
            r1.x = r0.w * ev;
            r1.y = dtv;
            // the coefficients of the vector-scale bound (header): c_j and a_j of this step; they involve s and rad of step j + 1 (the
            // lane above, or the chunk built before this one)
            float s_up = __shfl_down(r0.x, 1, 64), rad_up = __shfl_down(r1.x, 1, 64);
            if (lane == PCH - 1) { s_up = qq ? s_ab1 : s_ab0; rad_up = qq ? rad_ab1 : rad_ab0; }
            if (qq) { s_ab1 = rdl(r0.x, 0); rad_ab1 = rdl(r1.x, 0); } else { s_ab0 = rdl(r0.x, 0); rad_ab0 = rdl(r1.x, 0); }
            r1.z = fabsf(r0.w) * Hinf * (1.001f * sqrtf(fmaxf(nv, 1e-12f))) + fabsf(rad_up) * r0.z * r0.y;      // c_j >= |c3_j|
            r1.w = r0.y * fmaf(fabsf(s_up), gr1, gr0);                                                        // a_j
            TB.row[w][cj & 1][lane][qq][0] = r0;
            TB.row[w][cj & 1][lane][qq][1] = r1;
            if (in && w == 0 && (qq == 0 || two)) accA += zbar * ex;
        }
    };
    chunk_rows((N - 1) / PCH);
    if ((N - 1) / PCH > 0 && ((N - 1) & (PCH - 1)) == 0) chunk_rows((N - 1) / PCH - 1);
    {
        const int cl = (N - 1) / RCH;
        rho_stage<PD>(P, RS, cl, cl & 1, 64 * w + lane);
        if (cl > 0) rho_stage<PD>(P, RS, cl - 1, (cl - 1) & 1, 64 * w + lane);
    }
    if (threadIdx.x < 32) reinterpret_cast<unsigned*>(ymx_tab)[threadIdx.x] = 0u;
    __syncthreads();

    f2 ga = f2{0.f, 0.f}, gb = ga;                      // g: cotangent of u_{k+1}, (own component, the partner's: MFMA register 1)
    f2 una = ga, unb = ga;                              // u_{k+1} = rho_k yhat_k, (own, partner)
    float facca = 0.f, faccb = 0.f, accS = 0.f;
    unsigned ymxb = 0u;                                 // max |ybar| of this lane over the clip (float bits), for the gradient GEMM's scale
    float sda = 0.f, sdb = 0.f;                         // ((Q + s R^dagger) ybar) rows of the step before: its Abar term is added a step late
    u4 o0, o1;                                          // the wave's own K-steps: [K half] (both pieces stacked in the A rows)
    float4 rh, rhp;
    f4 S0, S1, SP0, SP1;
    float c3a, c3b;                                     // te_k (H y_k) - ok_k yhat_k rad_{k+1} inv_k
    float sS = 1.f, iS = 1.f;                           // the vector's scale of the step about to run, and its inverse
    auto rho_rows = [&](int k) { return *reinterpret_cast<const float4*>(&RS.row[(k / RCH) & 1][k & (RCH - 1)][ia]); };
    auto tab_row = [&](int k, int half) { return TB.row[w][(k / PCH) & 1][k & (PCH - 1)][q][half]; };
    typedef unsigned u2b __attribute__((ext_vector_type(2)));
    auto row_at = [&](int k) {
        const int kc = k > 0 ? k : 0;
        const u2b y = __builtin_amdgcn_raw_buffer_load_b64(rs_st, voff, kc * (8 * PD * 4), 0);
        const u2b h = __builtin_amdgcn_raw_buffer_load_b64(rs_st, voff + 4 * PD * 4, kc * (8 * PD * 4), 0);
        return make_float4(__uint_as_float(y.x), __uint_as_float(y.y), __uint_as_float(h.x), __uint_as_float(h.y));
    };
    float4 ring0, ring1, ring2, ring3, ring4, ring5, ring6, ring7;
    {
        const int k0 = N - 1;
        ring0 = row_at(k0 - ((k0 - 0) & 7));
        ring1 = row_at(k0 - ((k0 - 1) & 7));
        ring2 = row_at(k0 - ((k0 - 2) & 7));
        ring3 = row_at(k0 - ((k0 - 3) & 7));
        ring4 = row_at(k0 - ((k0 - 4) & 7));
        ring5 = row_at(k0 - ((k0 - 5) & 7));
        ring6 = row_at(k0 - ((k0 - 6) & 7));
        ring7 = row_at(k0 - ((k0 - 7) & 7));
        const int k1 = k0 > 0 ? k0 - 1 : 0;
        rh = rho_rows(k0);
        S0 = tab_row(k0, 0);
        S1 = tab_row(k0, 1);
        rhp = rho_rows(k1);
        SP0 = tab_row(k1, 0);
        SP1 = tab_row(k1, 1);
        const float4 cur = row_at(k0);
        c3a = S0.w * cur.z;                             // rad_N = 0
        c3b = S0.w * cur.w;
        sS = gg::pow2_scale(S1.z, 15);                  // ybar_{N-1} = c3_{N-1} (g = 0): its bound c_{N-1} (rad_N = 0)
        iS = __uint_as_float(0x7F000000u - __float_as_uint(sS));
    }
    const float2 psa = P.psi0[ia], psb = P.psi0[ib];
    const f2 ps0a = odd ? f2{psa.y, psa.x} : f2{psa.x, psa.y}, ps0b = odd ? f2{psb.y, psb.x} : f2{psb.x, psb.y};   // u_0 = psi_0 (own, partner)
    const unsigned a_tab = lds_addr_of(&TB.row[w][0][0][q][0]);
    const unsigned a_ym = lds_addr_of(&ymx_tab[0][q][0][0]);        // + 64 parity: this clip's eight entries
    auto write_image = [&](int par, float xa, float xb) {
        unsigned hi, lo;
        split_f16x2(xa, xb, hi, lo);
        unsigned char* b0p = L.vec[par][0];
        unsigned char* b1p = L.vec[par][1];
        *reinterpret_cast<unsigned*>(b0p + g.wr1) = hi;
        *reinterpret_cast<unsigned*>(b0p + g.wr2) = hi ^ 0x80008000u;
        *reinterpret_cast<unsigned*>(b1p + g.wr2) = lo;               // piece 1: array kind k in the place of kind 3 - k (chain_lane)
        *reinterpret_cast<unsigned*>(b1p + g.wr1) = lo ^ 0x80008000u;
    };
#if defined(CMPS_DIAG) && defined(C16_TIMING)              // diagnostic builds only: s_memtime stamps of a step's phases
    unsigned long long tS[7] = {0, 0, 0, 0, 0, 0, 0}, tAcc[6] = {0, 0, 0, 0, 0, 0}, tN = 0;
#define C16_STAMP(i) tS[i] = __builtin_readcyclecounter();
#define C16_STAMP_DEP(i, x) { asm volatile("" : "+v"(x)); tS[i] = __builtin_readcyclecounter(); }
#define C16_STAMPS_END() { for (int z = 0; z < 6; ++z) tAcc[z] += tS[z + 1] - tS[z]; ++tN; }
#else
#define C16_STAMP(i)
#define C16_STAMP_DEP(i, x)
#define C16_STAMPS_END()
#endif
#define QL_(i) FQl[QLITE ? 0 : (i)]
    // one step; J = k & 7 (static: ring slots and the image parity).  The chain is  g -> conj(rho) g -> ybar -> scaled fp16 pieces (own K
    // ranges read back) -> [barrier] -> K-steps -> g; the slots behind the MFMA pairs carry the step's LDS reads (two K-steps ahead of
    // their use, k_fwd_chain16) and the off-chain work of k_bwd_pair's pieces.
#define C16B_STEP(J, CUR, PRV, COND)                                                                                               \
    if (COND) {                                                                                                                    \
        constexpr int p = (J) & 1;                                                                                                 \
        const int k = 8 * blk + (J);                                                                                               \
        const int km2 = k > 1 ? k - 2 : 0;                                                                                         \
        if ((k & (PCH - 1)) == 1 && k > 1) chunk_rows(k / PCH - 1);                                                                \
        if ((k & (RCH - 1)) == RCH - 1 && k != N - 1 && k >= RCH)                                                                  \
            rho_stage<PD>(P, RS, k / RCH - 1, (k / RCH - 1) & 1, 64 * w + lane);                                                   \
        C16_STAMP(0)                                                                                                               \
        /* ---- the chain ---- */                                                                                                  \
        const float hba = rh.x * ga.x - sgn * rh.y * ga.y;           /* conj(rho_k) g */                                           \
        const float hbb = rh.z * gb.x - sgn * rh.w * gb.y;                                                                         \
        const float yba = fmaf(hba, S0.y, c3a), ybb = fmaf(hbb, S0.y, c3b);                                                        \
        write_image(p, yba * sS, ybb * sS);                                                                                        \
        rd_own<p * 2 * VEC>(g.lo[0], g.hi[0], o0, o1);                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        C16_STAMP(1)                                                                                                               \
        /* ---- behind the stores: ybar out (for the gradient GEMM) ---- */                                                        \
        __builtin_amdgcn_raw_buffer_store_b64(u2b{__float_as_uint(yba), __float_as_uint(ybb)}, rs_yb, voff, k * (4 * PD * 4), 0);   \
        const f2 una_ = una, unb_ = unb, ga_ = ga, gb_ = gb;                                                                       \
        unsigned mbits = max(__float_as_uint(yba) & 0x7FFFFFFFu, __float_as_uint(ybb) & 0x7FFFFFFFu);     /* max |ybar| as float bits */ \
        __builtin_amdgcn_sched_barrier(0);                                                                                         \
        C16_STAMP_DEP(2, mbits)                                                                                                    \
        lds_barrier();                                                                                                             \
        C16_STAMP(3)                                                                                                               \
        f4 cQ0, cQ1, cD0, cD1, nS0, nS1;                                                                                           \
        u4 v[2 * KH - 2];                                                                                                          \
        float4 nrh;                                                                                                                \
        f2 uka, ukb;                                                                                                               \
        float ypa, ypb, pypa, pypb, pyba, pybb, sSn;                                                                         \
        u4 ym[2];                                                                                                                  \
        const unsigned ax0 = a_tab + 64 * (((km2 / PCH) & 1) * PCH + (km2 & (PCH - 1)));                                           \
        const float invp = SP0.y;                                                                                                  \
        auto slot = [&](auto kidx_c, auto sl_c) {                                                                                  \
            constexpr int kidx = decltype(kidx_c)::value, sl = decltype(sl_c)::value;                                              \
            if constexpr (kidx < NR && sl == 0) {                    /* operand of rest K-step kidx */                             \
                constexpr int half = kidx / (KH - 1), t = 1 + kidx % (KH - 1);                                                     \
                rd128<p * 2 * VEC>(half ? g.hi[t] : g.lo[t], v[kidx]);                                                             \
            }                                                                                                                      \
            if constexpr (kidx == NR && sl == 0) rd128<0>(ax0, nS0);  /* scalar rows of step k - 2 */                              \
            if constexpr (kidx == NR && sl == 2) rd128<16>(ax0, nS1);                                                              \
            /* the pieces (k_bwd_pair): one small group of VALU per slot, pinned so that it stays in its slot */                    \
            if constexpr (kidx == 0 && sl == 1) { ypa = PRV.x * invp; ypb = PRV.y * invp; PAIR_PIN2(ypa, ypb); }                   \
            if constexpr (kidx == 0 && sl == 3) { pypa = partner16(ypa, odd); PAIR_PIN1(pypa); }                                   \
            if constexpr (kidx == 0 && sl == 5) { pypb = partner16(ypb, odd); PAIR_PIN1(pypb); }                                   \
            if constexpr (kidx == 1 && sl == 1) {                                                                                  \
                const float ria = sgn * rhp.y;                                                                                     \
                uka = f2{rhp.x * ypa + ria * pypa, rhp.x * pypa - ria * ypa};                                                      \
                PAIR_PIN1(uka);                                                                                                    \
            }                                                                                                                      \
            if constexpr (kidx == 1 && sl == 3) {                                                                                  \
                const float rib = sgn * rhp.w;                                                                                     \
                ukb = f2{rhp.z * ypb + rib * pypb, rhp.z * pypb - rib * ypb};                                                      \
                PAIR_PIN1(ukb);                                                                                                    \
            }                                                                                                                      \
            if constexpr (kidx == 1 && sl == 5) {                    /* the g-independent part of ybar_{k-1} and its bound */      \
                const float radk = S1.x * SP0.z * invp;                                                                            \
                c3a = fmaf(SP0.w, PRV.z, -(ypa * radk));                                                                           \
                c3b = fmaf(SP0.w, PRV.w, -(ypb * radk));                                                                           \
                PAIR_PIN2(c3a, c3b);                                                                                               \
            }                                                                                                                      \
            if constexpr (kidx == 2 && sl == 1) { pyba = partner16(yba, odd); PAIR_PIN1(pyba); }             \
            if constexpr (kidx == 2 && sl == 3) { pybb = partner16(ybb, odd); PAIR_PIN1(pybb); }                        \
            if constexpr (kidx == 2 && sl == 5) { accS += sda * una_.x + sdb * unb_.x; PAIR_PIN1(accS); }               \
            if constexpr (kidx == 3 && sl == 1) CUR = row_at(k - 8);                                                    \
            if constexpr (kidx == 3 && sl == 3) nrh = rho_rows(km2);                                                    \
            /* the frequency gradient and the wave's max |ybar_k| (-> LDS, for the scale two steps on) */                            \
            if constexpr (kidx == 2 && sl == 4) {                                                                                  \
                facca += S1.y * (ga_.y * una_.x - ga_.x * una_.y);   /* dt_k Im(g conj(u_{k+1})) (meaningful in the Re lanes) */   \
                faccb += S1.y * (gb_.y * unb_.x - gb_.x * unb_.y);                                                                 \
                PAIR_PIN2(facca, faccb);                                                                                           \
            }                                                                                                                      \
            if constexpr (kidx == 3 && sl == 4) {                                                                                  \
                ymxb = max(ymxb, mbits);                                                                                           \
                mbits = max(mbits, dpp_movu<0x128>(mbits)); mbits = max(mbits, dpp_movu<0x124>(mbits));                            \
                PAIR_PIN1(mbits);                                                                                                  \
            }                                                                                                                      \
            if constexpr (kidx == 3 && sl == 5) {                                                                                  \
                mbits = max(mbits, dpp_movu<0x122>(mbits)); mbits = max(mbits, dpp_movu<0x121>(mbits));                            \
                if ((lane & 15) == 0) ymx_tab[p][q][w][odd ? 1 : 0] = mbits;                                                       \
            }                                                                                                                      \
            /* max |ybar_{k+1}| of this lane's clip over the waves (left by the step before, published by this step's barrier) */      \
            if constexpr (kidx == 0 && sl == 4) rd128<0>(a_ym + 64 * (p ^ 1), ym[0]);                                              \
            if constexpr (kidx == 0 && sl == 5) rd128<16>(a_ym + 64 * (p ^ 1), ym[1]);                                             \
            if constexpr (kidx == NR + 1 && sl == 1) {               /* the scale of ybar_{k-1}'s image: the header's bound from Y_{k+1}       \
                                                                        (everything older than the two table rows has landed: this K-step's counted wait) */ \
                asm volatile("" : "+v"(ym[0]), "+v"(ym[1]));                                                                       \
                unsigned mb = max(max(ym[0].x, ym[0].y), max(ym[0].z, ym[0].w));                                                   \
                if constexpr (PWV > 2) mb = max(mb, max(ym[1].x, ym[1].y));                                                        \
                if constexpr (PWV > 3) mb = max(mb, max(ym[1].z, ym[1].w));                                                        \
                const float bnd = fmaf(SP1.w, fmaf(S1.w, __uint_as_float(mb), S1.z), SP1.z);    /* a_{k-1} (a_k Y_{k+1} + c_k) + c_{k-1} */ \
                sSn = gg::pow2_scale(bnd, 15);                                                                                     \
                PAIR_PIN1(sSn);                                                                                                    \
            }                                                                                                                      \
        };                                                                                                                         \
        kstep_st<15, true, QLITE, true>(FDh[0], FDl[0], FDh[KS], FDl[KS], FQh[0], QL_(0), FQh[KS], QL_(KS), o0, cD0, cD1, cQ0, cQ1,          \
                                        [&](auto sl) { slot(ic<0>{}, sl); });                                                      \
        kstep_st<15, false, QLITE, true>(FDh[KH], FDl[KH], FDh[KS + KH], FDl[KS + KH], FQh[KH], QL_(KH), FQh[KS + KH], QL_(KS + KH), o1,   \
                                         cD0, cD1, cQ0, cQ1, [&](auto sl) { slot(ic<1>{}, sl); });                                 \
        gg::static_for<0, NR>([&](auto ic_) {                                                                                      \
            constexpr int I = decltype(ic_)::value;                                                                                \
            constexpr int T_ = I < KH - 1 ? 1 + I : KH + 1 + (I - (KH - 1));                                                       \
            /* in flight behind this K-step's operand: what the K-step before issued -- the next operand (behind the first one also   \
               the two ymx rows), or the two table rows */                                                                        \
            kstep_st<(I == 0 ? 3 : I + 1 < NR ? 1 : 2), false, QLITE, true>(FDh[T_], FDl[T_], FDh[KS + T_], FDl[KS + T_], FQh[T_], QL_(T_),     \
                                                              FQh[KS + T_], QL_(KS + T_), v[I], cD0, cD1, cQ0, cQ1,               \
                                                              [&](auto sl) { slot(ic<I + 2>{}, sl); });                            \
        });                                                                                                                        \
        C16_STAMP(4)                                                                                                               \
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(nS0), "+v"(nS1) :: "memory");                                                   \
        C16_STAMP_DEP(5, cQ1[1])                                                                                                   \
        if (k == 0) { uka = ps0a; ukb = ps0b; }                      /* u_0 = psi_0 */                                             \
        {   /* g = ybar + (Q + s R^dagger) ybar, (own, partner) pairs; the accumulators carry the scales sQ sS and sD sS */           \
            const float cq = iQ * iS, cd = S0.x * (iD * iS);                                                                       \
            const f2 da = st_sum(cQ0) * cq + cd * st_sum(cD0);                                                                     \
            const f2 db = st_sum(cQ1) * cq + cd * st_sum(cD1);                                                                     \
            ga = f2{yba, pyba} + da;                                                                                               \
            gb = f2{ybb, pybb} + db;                                                                                               \
            sda = da.x; sdb = db.x;                                                                                                \
        }                                                                                                                          \
        sS = sSn;                                                                                                                  \
        iS = __uint_as_float(0x7F000000u - __float_as_uint(sSn));    /* 1 / sS: exact for a power of two */                        \
        una = uka; unb = ukb;                                                                                                      \
        rh = rhp; S0 = SP0; S1 = SP1;                                                                                              \
        rhp = nrh; SP0 = nS0; SP1 = nS1;                                                                                           \
        C16_STAMP_DEP(6, ga)                                                                                                       \
        C16_STAMPS_END()                                                                                                           \
    }

    int blk = (N + 7) / 8 - 1;
    if (N & 7) {
        C16B_STEP(7, ring7, ring6, 8 * blk + 7 < N)
        C16B_STEP(6, ring6, ring5, 8 * blk + 6 < N)
        C16B_STEP(5, ring5, ring4, 8 * blk + 5 < N)
        C16B_STEP(4, ring4, ring3, 8 * blk + 4 < N)
        C16B_STEP(3, ring3, ring2, 8 * blk + 3 < N)
        C16B_STEP(2, ring2, ring1, 8 * blk + 2 < N)
        C16B_STEP(1, ring1, ring0, 8 * blk + 1 < N)
        C16B_STEP(0, ring0, ring7, true)
        --blk;
    }
    for (; blk >= 0; --blk) {
        C16B_STEP(7, ring7, ring6, true)
        C16B_STEP(6, ring6, ring5, true)
        C16B_STEP(5, ring5, ring4, true)
        C16B_STEP(4, ring4, ring3, true)
        C16B_STEP(3, ring3, ring2, true)
        C16B_STEP(2, ring2, ring1, true)
        C16B_STEP(1, ring1, ring0, true)
        C16B_STEP(0, ring0, ring7, true)
    }
#undef C16B_STEP
#undef QL_
#if defined(CMPS_DIAG) && defined(C16_TIMING)
    if (blockIdx.x == 0 && lane == 0)
        printf("k_bwd_chain16 wave %d, cycles per step: chain VALU + image stores + own reads issued %.1f | behind the stores %.1f | wait + barrier %.1f | "
               "K-steps issued %.1f | tables + accumulators ready %.1f | tail %.1f\n", w, (double)tAcc[0] / tN, (double)tAcc[1] / tN, (double)tAcc[2] / tN,
               (double)tAcc[3] / tN, (double)tAcc[4] / tN, (double)tAcc[5] / tN);
#endif
    accS += sda * una.x + sdb * unb.x;                                 // the Abar term of step 0
    // ---- the pair's slab: f | psi0bar_re | psi0bar_im | A (as k_bwd_wide; the R / Q sections are written by k_grad_gemm) ----
    float* slab = P.slabs + (size_t)blockIdx.x * P.slab_floats;
    const int DD = PD * PD;
    {
        const float fta = both_clips(facca * wq), ftb = both_clips(faccb * wq);
        const float gta = both_clips(ga.x * wq), gtb = both_clips(gb.x * wq);
        if (g.f == 0) {
            slab[4 * DD + ia] = fta;
            slab[4 * DD + ib] = ftb;
            slab[4 * DD + PD + ia] = gta;
            slab[4 * DD + PD + ib] = gtb;
        } else if (g.f == 1) {
            slab[4 * DD + 2 * PD + ia] = gta;
            slab[4 * DD + 2 * PD + ib] = gtb;
        }
    }
    {
        float t = accS * wq, a = (w == 0 ? accA : 0.f);
        float ymx = __uint_as_float(ymxb);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { t += __shfl_xor(t, off, 64); a += __shfl_xor(a, off, 64); ymx = fmaxf(ymx, __shfl_xor(ymx, off, 64)); }
        if (lane == 0) { redA[w] = -(a / (A * A)) - t / A; L.red[w][0] = ymx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f, m = 0.f;
#pragma unroll
            for (int i = 0; i < PWV; ++i) { tot += redA[i]; m = fmaxf(m, L.red[i][0]); }
            slab[4 * DD + 3 * PD] = tot;
            slab[4 * DD + 3 * PD + 1] = 0.f;
            P.opmax[blockIdx.x] = m;
        }
    }
}

#undef C16_STAMP
#undef C16_STAMP_DEP
#undef C16_STAMPS_END

hipError_t launch_bwd_chain16(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    if (P.DP == 128) {
        hipLaunchKernelGGL((k_bwd_chain16<128, true>), dim3(nb), dim3(256), 0, s, P, audio);
        hipLaunchKernelGGL((k_bwd_chain16<128, false>), dim3(nb), dim3(256), 0, s, P, audio);
    } else if (P.DP == 96) {
        hipLaunchKernelGGL((k_bwd_chain16<96, true>), dim3(nb), dim3(192), 0, s, P, audio);
        hipLaunchKernelGGL((k_bwd_chain16<96, false>), dim3(nb), dim3(192), 0, s, P, audio);
    } else if (P.DP == 64) {
        hipLaunchKernelGGL((k_bwd_chain16<64, true>), dim3(nb), dim3(128), 0, s, P, audio);
        hipLaunchKernelGGL((k_bwd_chain16<64, false>), dim3(nb), dim3(128), 0, s, P, audio);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_bwd_pair(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    if (P.DP == 128) hipLaunchKernelGGL(k_bwd_pair<128>, dim3(nb), dim3(256), 0, s, P, audio);
    else if (P.DP == 96) hipLaunchKernelGGL(k_bwd_pair<96>, dim3(nb), dim3(192), 0, s, P, audio);
    else if (P.DP == 64) hipLaunchKernelGGL(k_bwd_pair<64>, dim3(nb), dim3(128), 0, s, P, audio);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace cmps
namespace cmps {

// ------------------------------------------------------------------------------------------------
// gradient contraction: k_grad_gemm (cmps_grad_gemm.h) with one bf16 piece per operand (each of te y | s ybar | ybar | y | u rounded
// to bf16 once: the rounding points of oracle/cmps_oracle.py::psi_bf16_scan) on this family's rows -- [clip][component][row] inside a
// step's y / ybar vector, 1 / sqrtf normalisation as in k_bwd_pair
// ------------------------------------------------------------------------------------------------
template <int PD>
struct PairRows {
    static __device__ __forceinline__ int y_off(int tid, int c) { return ((((tid >> 3) & 1) * 2 + c) * PD) + 8 * (tid >> 4) + (tid & 7); }
    static __device__ __forceinline__ int yb_off(int tid, int c) { return y_off(tid, c); }
    static __device__ __forceinline__ float rsq(float m) { return 1.0f / sqrtf(m); }
};

hipError_t launch_grad_pair(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    if (P.DP == 128) hipLaunchKernelGGL((k_grad_gemm<128, 1, PairRows<128>>), dim3(nb), dim3(256), 0, s, P, audio);
    else if (P.DP == 96) hipLaunchKernelGGL((k_grad_gemm<96, 1, PairRows<96>>), dim3(nb), dim3(192), 0, s, P, audio);
    else if (P.DP == 64) hipLaunchKernelGGL((k_grad_gemm<64, 1, PairRows<64>>), dim3(nb), dim3(128), 0, s, P, audio);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace cmps
