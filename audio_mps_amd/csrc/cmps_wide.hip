// 32 < D <= 128 in float32 ("wide" kernels, CMPS_VARIANT_WIDE; what AUTO selects above D = 32).
//
// The reference computes this path in float32 / complex64 at every bond dimension (model.py:300-325); the bf16 pair
// kernels (cmps_pair.hip) are an opt-in trade of accuracy for speed, and the block kernels (cmps_block.hip) re-read the
// matrices from L2 every step.  These kernels keep the float32 arithmetic and make the matrices resident:
//
//   workgroup = one PAIR of clips, PD / 16 waves (PD = D padded to 64 / 96 / 128): two waves per SIMD at PD = 128.
//   wave w owns rows 16 w .. 16 w + 15 of every matrix; lane = 8 q + i:
//     mat-vec:    rows 16 w + i and 16 w + 8 + i, K slice q = columns q PD/8 .. (q + 1) PD/8 - 1; the two matrices of a scan
//                 (R and Q forward, R^dagger and Q in the reverse scan) are 4 PD/8 float2 = PD/2 .. 128 VGPRs per lane;
//     after it:   three halving stages (v_permlane32_swap, v_permlane16_swap, DPP row_ror:8) sum the eight K slices and
//                 leave ONE float per lane: (row 16 w + 8 (q >> 2) + i, component (q >> 1) & 1, clip q & 1).
//   The two clips of a pair ride in the two halves of v_pk_fma_f32: a broadcast vector lives in LDS as one float4
//   (re clip0, re clip1, im clip0, im clip1) per component, a matrix entry is a (re, im) register pair, and the complex
//   multiply-accumulate of both clips is four packed FMAs with op_sel broadcasts -- no swizzles.  The merged matrix
//   M_k = Q + s_k R (s_k differs per clip) is formed on the fly, two packed FMAs per entry: 6 instructions per entry
//   and clip pair instead of 8.
//   H = R + R^dagger (the loss product H y of the forward) does not fit next to R and Q in the 256 architectural
//   VGPRs a wave can address (AGPRs are no VALU operands), so it sits in LDS in lane order (PD^2 * 8 B: 128 KB at PD = 128).
//
// Forward  (k_fwd_wide):  one LDS-only barrier per step; the loss product of step k - 1 shares step k's barrier interval.
// Reverse  (k_bwd_wide):  the cotangent recursion g -> conj(rho) g -> ybar -> (Q + s R^dagger) ybar, analytic radial
//                         derivative (see cmps_pair.hip); writes ybar_k (float32, lane order) for the gradient kernel.
// Gradient (k_grad_gemm, cmps_grad_gemm.h): Rbar = sum (te y) y^dagger + (s ybar) u^dagger, Qbar = sum ybar u^dagger as
//                         v_mfma_f32_32x32x16_bf16 GEMMs over K = (clip, step, {re, im}); every operand is split EXACTLY into
//                         three bf16 pieces (8 + 8 + 8 significand bits) on the fly and the six significant piece products
//                         are accumulated in fp32 (24 operand bits: what is dropped is <= 2^-23 |a||b|), the same
//                         fp32-faithful product as CMPS_RANK1_BF16X3 of the D <= 32 kernels; CMPS_RANK1_BF16X2 keeps two
//                         pieces / three products (16 operand bits).
// Stash (Dev::stash, layout 4): [pair][step][y | H y][wave][lane] float32 (4 PD floats per vector: the lane order above);
// ybar: the same vector shape per (pair, step) in Dev::gops.
#include "cmps_internal.h"
#include "cmps_grad_gemm.h"

namespace cmps {

namespace {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u4w __attribute__((ext_vector_type(4)));
typedef unsigned u2w __attribute__((ext_vector_type(2)));
typedef short bf8w __attribute__((ext_vector_type(8)));
typedef float f16w __attribute__((ext_vector_type(16)));

constexpr int WCH = 64;      // steps per chunk of per-step scalars (one step per lane)

__device__ __forceinline__ v2f mkv2(float a, float b) { v2f r; r.x = a; r.y = b; return r; }
__device__ __forceinline__ v2f lo_of(v4f q) { return __builtin_shufflevector(q, q, 0, 1); }
__device__ __forceinline__ v2f hi_of(v4f q) { return __builtin_shufflevector(q, q, 2, 3); }
__device__ __forceinline__ float wrdl(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
template <int CTRL>
__device__ __forceinline__ float wdpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// lanes l and l ^ 32: lower lanes receive x + x', upper lanes y + y'
__device__ __forceinline__ float swap32_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// lanes l and l ^ 16: even 16-lane rows receive x + x', odd rows y + y'
__device__ __forceinline__ float swap16_add(float x, float y) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(y), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the value of lane l ^ 16
__device__ __forceinline__ float partner16(float x, bool odd_row) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return odd_row ? __uint_as_float(r[0]) : __uint_as_float(r[1]);
}
// partial sums of the eight K slices (two rows x {re, im} x packed clips) -> this lane's own (row, component, clip)
__device__ __forceinline__ float reduce_slices(v2f re0, v2f im0, v2f re1, v2f im1, bool clip1) {
    const float tr0 = swap32_add(re0.x, re1.x), tr1 = swap32_add(re0.y, re1.y);       // row select (q bit 2)
    const float ti0 = swap32_add(im0.x, im1.x), ti1 = swap32_add(im0.y, im1.y);
    const float c0 = swap16_add(tr0, ti0), c1 = swap16_add(tr1, ti1);                 // component select (q bit 1)
    const float keep = clip1 ? c1 : c0, give = clip1 ? c0 : c1;                       // clip select (q bit 0)
    return keep + wdpp<0x128>(give);                                                  // row_ror:8: lane l ^ 8
}
// sum over the lanes of this wave that carry the same clip (q & 1); every lane receives its clip's total
__device__ __forceinline__ float clip_wave_sum(float x) {
    x += wdpp<0xB1>(x);           // quad_perm [1,0,3,2]
    x += wdpp<0x4E>(x);           // quad_perm [2,3,0,1]
    x += wdpp<0x141>(x);          // row_half_mirror: the other quad of this 8-lane group
    x = swap16_add(x, x);
    x = swap32_add(x, x);
    return x;
}
__device__ __forceinline__ float rsq_newton(float m) {   // 1 / sqrt(m): v_rsq_f32 + one Newton step (same in all three kernels)
    const float r = __builtin_amdgcn_rsqf(m);
    return r * (1.5f - 0.5f * m * r * r);
}
__device__ __forceinline__ void wide_barrier() {          // orders LDS traffic only (__syncthreads() would also drain the stash stores)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// acc(two rows) += (Q + s R)[row][col] * x[col] for both clips:  s2 = (s clip0, s clip1) in SGPRs, x = (re c0, re c1 | im c0, im c1)
__device__ __forceinline__ void col_merged(v2f& aRe0, v2f& aIm0, v2f& aRe1, v2f& aIm1, v2f r0, v2f q0, v2f r1, v2f q1, v2f s2,
                                           v2f xre, v2f xim) {
    v2f t0, t1, t2, t3;
    asm("v_pk_fma_f32 %4, %8, %12, %9 op_sel_hi:[0,1,0]\n\t"              // m_re = R_re s + Q_re   (per clip)
        "v_pk_fma_f32 %5, %8, %12, %9 op_sel:[1,0,1]\n\t"                 // m_im = R_im s + Q_im
        "v_pk_fma_f32 %6, %10, %12, %11 op_sel_hi:[0,1,0]\n\t"
        "v_pk_fma_f32 %7, %10, %12, %11 op_sel:[1,0,1]\n\t"
        "v_pk_fma_f32 %0, %4, %13, %0\n\t"                                // Re += m_re x_re
        "v_pk_fma_f32 %2, %6, %13, %2\n\t"
        "v_pk_fma_f32 %1, %4, %14, %1\n\t"                                // Im += m_re x_im
        "v_pk_fma_f32 %3, %6, %14, %3\n\t"
        "v_pk_fma_f32 %0, %5, %14, %0 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"  // Re -= m_im x_im
        "v_pk_fma_f32 %2, %7, %14, %2 neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %5, %13, %1\n\t"                                // Im += m_im x_re
        "v_pk_fma_f32 %3, %7, %13, %3"
        : "+v"(aRe0), "+v"(aIm0), "+v"(aRe1), "+v"(aIm1), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(r0), "v"(q0), "v"(r1), "v"(q1), "s"(s2), "v"(xre), "v"(xim));
}
// accR(two rows) += R[row][col] x[col], accQ(two rows) += Q[row][col] x[col] for both clips, unmerged (the sampler needs R ut by
// itself: the increment it scales R ut with depends on <R>)
__device__ __forceinline__ void col_two(v2f& rRe0, v2f& rIm0, v2f& rRe1, v2f& rIm1, v2f& qRe0, v2f& qIm0, v2f& qRe1, v2f& qIm1,
                                        v2f r0, v2f q0, v2f r1, v2f q1, v2f xre, v2f xim) {
#define WIDE_CMAC(RE, IM, M)                                                                            \
    "v_pk_fma_f32 %" #RE ", %" #M ", %12, %" #RE " op_sel_hi:[0,1,1]\n\t"                                  \
    "v_pk_fma_f32 %" #IM ", %" #M ", %13, %" #IM " op_sel_hi:[0,1,1]\n\t"                                  \
    "v_pk_fma_f32 %" #RE ", %" #M ", %13, %" #RE " op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"       \
    "v_pk_fma_f32 %" #IM ", %" #M ", %12, %" #IM " op_sel:[1,0,0]\n\t"
    asm(WIDE_CMAC(0, 1, 8) WIDE_CMAC(4, 5, 9) WIDE_CMAC(2, 3, 10) WIDE_CMAC(6, 7, 11)
        : "+v"(rRe0), "+v"(rIm0), "+v"(rRe1), "+v"(rIm1), "+v"(qRe0), "+v"(qIm0), "+v"(qRe1), "+v"(qIm1)
        : "v"(r0), "v"(q0), "v"(r1), "v"(q1), "v"(xre), "v"(xim));
#undef WIDE_CMAC
}
// acc(two rows) += M[row][col] * x[col] for both packed vectors, ONE matrix for both (k_fwd_wide_rho: the two vectors are columns of the
// same clip, so M_k = Q + s_k R is the same for both and is formed once per step, not per column pair): m = (re, im)
__device__ __forceinline__ void col_single(v2f& aRe0, v2f& aIm0, v2f& aRe1, v2f& aIm1, v2f m0, v2f m1, v2f xre, v2f xim) {
    asm("v_pk_fma_f32 %0, %4, %6, %0 op_sel_hi:[0,1,1]\n\t"                                    // Re += m_re x_re
        "v_pk_fma_f32 %2, %5, %6, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %4, %7, %1 op_sel_hi:[0,1,1]\n\t"                                    // Im += m_re x_im
        "v_pk_fma_f32 %3, %5, %7, %3 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %4, %7, %0 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"         // Re -= m_im x_im
        "v_pk_fma_f32 %2, %5, %7, %2 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %4, %6, %1 op_sel:[1,0,0]\n\t"                                       // Im += m_im x_re
        "v_pk_fma_f32 %3, %5, %6, %3 op_sel:[1,0,0]"
        : "+v"(aRe0), "+v"(aIm0), "+v"(aRe1), "+v"(aIm1)
        : "v"(m0), "v"(m1), "v"(xre), "v"(xim));
}
// acc(two rows) += H[row][col .. col + 1] * y[col .. col + 1] for both clips; h = (re, im, re, im) of two adjacent columns
__device__ __forceinline__ void col_pair_plain(v2f& aRe0, v2f& aIm0, v2f& aRe1, v2f& aIm1, v4f h0, v4f h1, v4f ya, v4f yb) {
    asm("v_pk_fma_f32 %0, %4, %8, %0 op_sel_hi:[0,1,1]\n\t"                                    // Re += h_re y_re
        "v_pk_fma_f32 %2, %6, %8, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %4, %9, %1 op_sel_hi:[0,1,1]\n\t"                                    // Im += h_re y_im
        "v_pk_fma_f32 %3, %6, %9, %3 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %4, %9, %0 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"         // Re -= h_im y_im
        "v_pk_fma_f32 %2, %6, %9, %2 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %4, %8, %1 op_sel:[1,0,0]\n\t"                                       // Im += h_im y_re
        "v_pk_fma_f32 %3, %6, %8, %3 op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %0, %5, %10, %0 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %2, %7, %10, %2 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %1, %5, %11, %1 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %3, %7, %11, %3 op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %5, %11, %0 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %2, %7, %11, %2 op_sel:[1,0,0] neg_lo:[1,0,0] neg_hi:[1,0,0]\n\t"
        "v_pk_fma_f32 %1, %5, %10, %1 op_sel:[1,0,0]\n\t"
        "v_pk_fma_f32 %3, %7, %10, %3 op_sel:[1,0,0]"
        : "+v"(aRe0), "+v"(aIm0), "+v"(aRe1), "+v"(aIm1)
        : "v"(lo_of(h0)), "v"(hi_of(h0)), "v"(lo_of(h1)), "v"(hi_of(h1)), "v"(lo_of(ya)), "v"(hi_of(ya)), "v"(lo_of(yb)),
          "v"(hi_of(yb)));
}

template <int PD>
struct WideGeom {
    static constexpr int NW = PD / 16;          // waves per workgroup
    static constexpr int NTH = 64 * NW;         // threads = 4 PD = (row, component, clip) positions of one vector
    static constexpr int KC = PD / 8;           // columns per lane
    static constexpr int VSL = KC + 1;          // float4 per K slice of a broadcast vector (one float4 of padding: bank spread)
    static constexpr int VEC4 = 8 * VSL;        // float4 per broadcast vector
    static constexpr size_t FWD_LDS = ((size_t)NW * KC * 64 + 4 * VEC4) * 16 + 2 * 2 * NW * 2 * 4;     // loss in the kernel
    static constexpr size_t FWD_LDS_CHAIN = ((size_t)2 * VEC4) * 16 + 2 * NW * 2 * 4;                     // chain only (SAVE)
};
// float4 index (and float offset inside it) of this lane's own value in a broadcast vector
template <int PD>
__device__ __forceinline__ int vec_float_index(int row, int comp, int clip) {
    constexpr int KC = WideGeom<PD>::KC, VSL = WideGeom<PD>::VSL;
    return ((row / KC) * VSL + row % KC) * 4 + comp * 2 + clip;
}
// float offset of the vector (pair, step, y / H y) in the stash
template <int PD>
__device__ __forceinline__ size_t wide_stash_vec(size_t pair, int N, int step, int yh) {
    return ((pair * N + step) * 2 + yh) * (size_t)(4 * PD);
}
template <int PD>
__device__ __forceinline__ size_t wide_ybar_vec(size_t pair, int N, int step) {
    return (pair * N + step) * (size_t)(4 * PD);
}


// x = hi + mid + lo exactly, each with 8 significant bits: returned as bf16 bit patterns in the upper halves
__device__ __forceinline__ void split3(float x, unsigned& h, unsigned& m, unsigned& l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2);                        // <= 8 significant bits: exact in bf16 (the low half is zero)
}
__device__ __forceinline__ unsigned pack_hi16(unsigned lo_word, unsigned hi_word) {   // (lo_word >> 16) | (hi_word & 0xFFFF0000)
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}
__device__ __forceinline__ bf8w piece_bits(u4w v, unsigned mask) {
    u4w t = {v.x ^ mask, v.y ^ mask, v.z ^ mask, v.w ^ mask};
    return __builtin_bit_cast(bf8w, t);
}


// per-step scalars of one clip from what the forward stashed (|y_k|^2, e_k) and the audio, in the reference's operation order
// (model.py:294, 303); shared by the reverse scan and the gradient kernel so that both see the same numbers
struct StepScal { float s, inv, ok, te, rad, zex; };
__device__ __forceinline__ StepScal step_scalars(float inc, float nv, float ev, float A) {
    StepScal r;
    const float ex = ev * inc;
    const float z = ex / A;
    const float zbar = -1.0f / (1.0f + z);
    r.s = inc / A;
    r.inv = rsq_newton(fmaxf(nv, 1e-12f));
    r.ok = nv > 1e-12f ? 1.f : 0.f;
    r.te = 2.0f * (zbar * inc / A);
    r.rad = r.te * ev;
    r.zex = zbar * ex;
    return r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------------------------------
// SAVE = false (loss only, no stash in the workspace): the loss product H y_{k-1} runs inside the kernel, on the VALU, H in LDS.
// SAVE = true (training): the kernel is the serial chain alone -- it stashes y_k and |y_k|^2 -- and H y, e_k = y_k . H y_k and
// the loss follow as a GEMM over all (clip, step) pairs at once (k_hy_wide, k_loss_wide below): the in-kernel product costs
// 128 of the 320 packed FMAs a wave issues per step (38 of 62 ms at configs[4]), the GEMM runs them on the matrix cores.
// LEGACY (round 5): the arithmetic of the previous-generation AudioMPS (cmps_legacy.hip has the recurrence and its
// graph.pbtxt lines) on the same kernels, as the D <= 32 wave kernels do it (cmps_wave2.hip).  With rho = 1 and psi_0 = e_0 in the tables
// (cmps_legacy_set_params) the chain is the same linear step y_k = inv_{k-1} (y_{k-1} + M_k y_{k-1}), M_k = Q + dt x_k R (Q a general
// complex matrix); what differs is scalar: e_k = (y_{k-1}^dagger H y_{k-1}) / max(|y_{k-1}|^2, 1e-12) on the normalised state BEFORE the
// update (e_0 = H_00), loss += (x_k - e_k)^2 / 2.
template <int PD, bool SAVE, bool LEGACY = false>
__global__ __launch_bounds__(4 * PD) void k_fwd_wide(Dev P, const float* __restrict__ audio, float* __restrict__ loss_out) {
    using G = WideGeom<PD>;
    constexpr int NW = G::NW, KC = G::KC, VSL = G::VSL, VEC4 = G::VEC4;
    constexpr bool LOSS = !SAVE;                                  // the loss product in this kernel
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
    v4f* Hl = reinterpret_cast<v4f*>(wide_lds);                   // [NW][KC][64]: H in lane order (LOSS only)
    v4f* uvec = LOSS ? Hl + NW * KC * 64 : Hl;                    // [2][VEC4]
    v4f* yvec = uvec + 2 * VEC4;                                  // [2][VEC4] (LOSS only)
    float* nrm = reinterpret_cast<float*>(LOSS ? yvec + 2 * VEC4 : uvec + 2 * VEC4);       // [2][NW][2]
    float* ee = nrm + 2 * NW * 2;                                 // [2][NW][2] (LOSS only)
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = lane >> 3, i = lane & 7;
    const int rowsel = q >> 2, comp = (q >> 1) & 1, clip = q & 1;
    const bool clip1 = clip != 0, im_lane = comp != 0;
    const int row = 16 * w + 8 * rowsel + i;
    const int N = P.N, T = P.T, NC = (N + WCH - 1) / WCH;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;   // an odd batch repeats its last clip (not stored)
    const bool two = b1 != b0;
    const float* xr0 = audio + (size_t)b0 * T;
    const float* xr1 = audio + (size_t)b1 * T;
    const float A = dev_A(P);

    // ---- matrices: rows 16 w + i (a) and 16 w + 8 + i (b), columns q KC .. ----
    v2f MR[2][KC], MQ[2][KC];
    {
        const int ra = 16 * w + i, rb = ra + 8, c0 = q * KC;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const float2 a = P.R[(size_t)ra * PD + c0 + j], b = P.R[(size_t)rb * PD + c0 + j];
            const float2 c = P.Q[(size_t)ra * PD + c0 + j], d = P.Q[(size_t)rb * PD + c0 + j];
            MR[0][j] = mkv2(a.x, a.y); MR[1][j] = mkv2(b.x, b.y);
            MQ[0][j] = mkv2(c.x, c.y); MQ[1][j] = mkv2(d.x, d.y);
        }
        // H = R + R^dagger in lane order: float4 number rs KC/2 + jp of this lane = H[row_rs][c0 + 2 jp], H[row_rs][c0 + 2 jp + 1]
#pragma unroll
        for (int rr = 0; LOSS && rr < KC; ++rr) {
            const int rs = rr / (KC / 2), jp = rr % (KC / 2), r_ = rs ? rb : ra, c_ = c0 + 2 * jp;
            const float2 x0 = P.R[(size_t)r_ * PD + c_], x1 = P.R[(size_t)r_ * PD + c_ + 1];
            const float2 t0 = P.RT[(size_t)r_ * PD + c_], t1 = P.RT[(size_t)r_ * PD + c_ + 1];   // RT[i][j] = R[j][i]
            Hl[(w * KC + rr) * 64 + lane] = v4f{x0.x + t0.x, x0.y - t0.y, x1.x + t1.x, x1.y - t1.y};
        }
    }
    const int own_f = vec_float_index<PD>(row, comp, clip);
    const int rd4 = q * VSL;                                      // first float4 of this lane's K slice
    float* stash = reinterpret_cast<float*>(P.stash);
    const size_t pos = (size_t)w * 64 + lane;

    const float2 p0 = P.psi0[row];
    float ut = im_lane ? p0.y : p0.x;                             // ut_0 = psi_0 (both clips)
    reinterpret_cast<float*>(uvec)[own_f] = ut;
    float yprev = 0.f;                                            // y_{k-1} of this lane (for e_{k-1} = y . H y)
    float sv0 = 0.f, sv1 = 0.f;                                   // s = x / A of the current 64 steps, lane <-> step
    float ebuf0 = 0.f, ebuf1 = 0.f, nbuf0 = 0.f, nbuf1 = 0.f;     // wave 0: e_k, |y_k|^2 of the current chunk, lane <-> step
    float loss0 = 0.f, loss1 = 0.f;
    float nq0 = 1.f, nq1 = 1.f;                                   // LEGACY loss in the kernel: |y_{k-2}|^2 (the sums of the iteration before)
    float fbel0 = 2.0f * P.R[0].x, fbel1 = fbel0, nbel0 = 1.f, nbel1 = 1.f;   // ... y^dagger H y and |y|^2 of the step below the chunk (psi_0 = e_0)
    float2 rho_next = P.rho[row];                                 // rho_0
    __syncthreads();

    for (int k = 0; k <= (LOSS ? N + 1 : N); ++k) {
        const int p = k & 1;
#if defined(CMPS_DIAG) && defined(WABL_NO_LOSSMV)    // diagnostic builds only (results are wrong): what the loss product costs
        const bool chain = k < N, lossmv = false;
#else
        const bool chain = k < N, lossmv = LOSS && k >= 1 && k <= N;
#endif
        if (chain && (k & (WCH - 1)) == 0) {                      // increments of the next 64 steps, one per lane (model.py:263, 303)
            const int idx = k + lane;
            const bool in0 = idx < T, in1 = idx + 1 < T;
            const float i0 = (in1 ? xr0[idx + 1] : 0.f) - (in0 ? xr0[idx] : 0.f), i1 = (in1 ? xr1[idx + 1] : 0.f) - (in0 ? xr1[idx] : 0.f);
            sv0 = LEGACY ? P.dt * i0 : i0 / A;
            sv1 = LEGACY ? P.dt * i1 : i1 / A;
        }
        const float2 rho_k = rho_next;
        // y_{k-1} goes out HERE, in front of the rho load: vector-memory operations retire in order, so waiting for rho_{k+1} at the
        // top of the next step then only waits for operations a full step old (with the store behind the load, every step waited
        // for its predecessor's store to be acknowledged)
        if (SAVE && k >= 1) stash[wide_stash_vec<PD>(blockIdx.x, N, k - 1, 0) + pos] = yprev;
        if (k + 1 < N) rho_next = P.rho[(size_t)(k + 1) * PD + row];
        // |y_{k-1}|^2 of both clips (published before the barrier of the previous iteration)
        float n0 = 1.f, n1 = 1.f;
        if (k >= 1) {
            n0 = n1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float2 t = *reinterpret_cast<const float2*>(&nrm[(p * NW + ww) * 2]);
                n0 += t.x; n1 += t.y;
            }
        }
        const float inv = k >= 1 ? rsq_newton(fmaxf(clip1 ? n1 : n0, 1e-12f)) : 1.f;      // model.py:332
        v2f cRe0 = mkv2(0.f, 0.f), cIm0 = cRe0, cRe1 = cRe0, cIm1 = cRe0;
        v2f hRe0 = cRe0, hIm0 = cRe0, hRe1 = cRe0, hIm1 = cRe0;
        if (chain) {
            const int kl = k & (WCH - 1);
            const v2f s2 = mkv2(wrdl(sv0, kl), wrdl(sv1, kl));
            const v4f* uv = uvec + p * VEC4 + rd4;
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                const v4f x = uv[j];
                col_merged(cRe0, cIm0, cRe1, cIm1, MR[0][j], MQ[0][j], MR[1][j], MQ[1][j], s2, lo_of(x), hi_of(x));
            }
        }
        if (lossmv) {                                             // H y_{k-1}
            const v4f* yv = yvec + p * VEC4 + rd4;
            const v4f* hl = Hl + (size_t)w * KC * 64 + lane;
#pragma unroll
            for (int jp = 0; jp < KC / 2; ++jp)
                col_pair_plain(hRe0, hIm0, hRe1, hIm1, hl[jp * 64], hl[(KC / 2 + jp) * 64], yv[2 * jp], yv[2 * jp + 1]);
        }
        if (chain) {
            const float acc = reduce_slices(cRe0, cIm0, cRe1, cIm1, clip1);
            const float y = inv * (ut + acc);                     // y_k = inv_{k-1} (ut + M_k ut)
            const float nn = clip_wave_sum(y * y);
            if (i == 0 && q < 2) nrm[((p ^ 1) * NW + w) * 2 + clip] = nn;
            if (LOSS) reinterpret_cast<float*>(yvec + (p ^ 1) * VEC4)[own_f] = y;
            const float py = partner16(y, im_lane);
            ut = rho_k.x * y + (im_lane ? rho_k.y : -rho_k.y) * py;      // ut_{k+1} = rho_k y_k (un-normalised)
            reinterpret_cast<float*>(uvec + (p ^ 1) * VEC4)[own_f] = ut;
            if (lossmv) {
                const float hy = reduce_slices(hRe0, hIm0, hRe1, hIm1, clip1);
                const float ep = clip_wave_sum(yprev * hy);
                if (i == 0 && q < 2) ee[((p ^ 1) * NW + w) * 2 + clip] = ep;
                if (SAVE) stash[wide_stash_vec<PD>(blockIdx.x, N, k - 1, 1) + pos] = hy;
            }
            yprev = y;
        } else if (lossmv) {                                      // k == N: the last loss product
            const float hy = reduce_slices(hRe0, hIm0, hRe1, hIm1, clip1);
            const float ep = clip_wave_sum(yprev * hy);
            if (i == 0 && q < 2) ee[((p ^ 1) * NW + w) * 2 + clip] = ep;
            if (SAVE) stash[wide_stash_vec<PD>(blockIdx.x, N, k - 1, 1) + pos] = hy;
        }
        if (w == 0 && LOSS) {                                     // bookkeeping: e_{k-2}, |y_{k-1}|^2, the loss (sequential float32)
            if (k >= 2) {
                const int ke = k - 2;
                float e0 = 0.f, e1 = 0.f;
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) {
                    const float2 t = *reinterpret_cast<const float2*>(&ee[(p * NW + ww) * 2]);
                    e0 += t.x; e1 += t.y;
                }
                if (lane == (ke & (WCH - 1))) { ebuf0 = e0; ebuf1 = e1; if (LEGACY) { nbuf0 = nq0; nbuf1 = nq1; } }
                if ((ke & (WCH - 1)) == WCH - 1 || ke == N - 1) {
                    const int c = ke / WCH, idx = c * WCH + lane;
                    const bool in = idx < N;
                    const float i0 = in ? xr0[idx + 1] - xr0[idx] : 0.f, i1 = in ? xr1[idx + 1] - xr1[idx] : 0.f;
                    float l0, l1;
                    if constexpr (LEGACY) {                       // the expectation on the normalised state of the step below (graph.pbtxt:11857-12819)
                        float f0 = __shfl_up(ebuf0, 1, 64), f1 = __shfl_up(ebuf1, 1, 64), m0 = __shfl_up(nbuf0, 1, 64), m1 = __shfl_up(nbuf1, 1, 64);
                        if (lane == 0) { f0 = fbel0; f1 = fbel1; m0 = nbel0; m1 = nbel1; }
                        fbel0 = wrdl(ebuf0, WCH - 1); fbel1 = wrdl(ebuf1, WCH - 1); nbel0 = wrdl(nbuf0, WCH - 1); nbel1 = wrdl(nbuf1, WCH - 1);
                        const float v0 = 1.0f / sqrtf(fmaxf(m0, 1e-12f)), v1 = 1.0f / sqrtf(fmaxf(m1, 1e-12f));
                        const float d0 = i0 - (f0 * v0) * v0, d1 = i1 - (f1 * v1) * v1;
                        l0 = in ? d0 * d0 / 2.0f : 0.f;
                        l1 = in ? d1 * d1 / 2.0f : 0.f;
                    } else {
                    l0 = in ? -logf(1.0f + (ebuf0 * i0) / A) : 0.f;        // model.py:294 operation order
                    l1 = in ? -logf(1.0f + (ebuf1 * i1) / A) : 0.f;
                    }
#pragma unroll
                    for (int j = 0; j < WCH; ++j) {               // model.py:279: sequential in time
                        loss0 += wrdl(l0, j);
                        loss1 += wrdl(l1, j);
                    }
                }
            }
        }
        if (LEGACY && LOSS) { nq0 = n0; nq1 = n1; }               // |y_{k-1}|^2: the e of step k - 1 is finished in the next iteration
        if (w == 0 && SAVE && k >= 1) {                           // |y_{k-1}|^2 rows of the scalar stash, one chunk per 64 steps
            const int kn = k - 1;
            if (lane == (kn & (WCH - 1))) { nbuf0 = n0; nbuf1 = n1; }
            if ((kn & (WCH - 1)) == WCH - 1 || kn == N - 1) {
                const int c = kn / WCH;
                if (c * WCH + lane < N) {
                    P.scal[((size_t)b0 * NC + c) * 128 + lane] = nbuf0;
                    if (two) P.scal[((size_t)b1 * NC + c) * 128 + lane] = nbuf1;
                }
            }
        }
        wide_barrier();
    }
    if (LOSS && w == 0 && lane == 0) {
        loss_out[b0] = loss0;
        if (two) loss_out[b1] = loss1;
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// RhoCMPS forward chain on the wide layout (round 5; model.py:133-144, 152-158, 172-203).  rho = sum_a phi_a phi_a^dagger is carried as
// its columns (cmps_rho.hip has the argument): every column takes the SAME linear step y_a = inv (ut_a + (Q + s R) ut_a), coupled only through
// the trace n = sum_a |y_a|^2 (inv = 1 / sqrt(n_{k-1})) and the expectation e = sum_a y_a^dagger H y_a.  One workgroup per clip, R and Q
// register resident as in k_fwd_wide; the two halves of v_pk_fma_f32 carry two COLUMNS of the clip instead of two clips, and a step loops
// over the clip's column pairs (broadcast vectors [2 buffers][pairs] in LDS), one LDS-only barrier per step.  It stashes y of the
// column pair cp of clip b as the row of VIRTUAL pair b pairs + cp -- the row format of k_fwd_wide -- so that H y (k_hy_wide), the reverse
// chain (k_bwd_wide: with the per-step scalars of the clip the column cotangents do not couple at all) and the gradient GEMM
// (k_grad_gemm) run unchanged on rank x as many "clips"; |y|^2 of the clip goes to the per-clip scalar rows (rscal).
// ------------------------------------------------------------------------------------------------------------------------
template <int PD>
__global__ __launch_bounds__(4 * PD) void k_fwd_wide_rho(Dev P, const float* __restrict__ audio, float* __restrict__ rscal, int npairs) {
    using G = WideGeom<PD>;
    constexpr int NW = G::NW, KC = G::KC, VSL = G::VSL, VEC4 = G::VEC4;
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
    v4f* uvec = reinterpret_cast<v4f*>(wide_lds);                 // [2][npairs][VEC4]
    float* nrm = reinterpret_cast<float*>(uvec + (size_t)2 * npairs * VEC4);   // [2][NW]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = lane >> 3, i = lane & 7;
    const int rowsel = q >> 2, comp = (q >> 1) & 1, clip = q & 1;   // clip = the column's parity inside its pair
    const bool clip1 = clip != 0, im_lane = comp != 0;
    const int row = 16 * w + 8 * rowsel + i;
    const int N = P.N, T = P.T, NC = (N + WCH - 1) / WCH;
    const int b = blockIdx.x;
    const float* xr = audio + (size_t)b * T;
    const float A = dev_A(P);
    v2f MR[2][KC], MQ[2][KC];
    {
        const int ra = 16 * w + i, rb = ra + 8, c0 = q * KC;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const float2 a = P.R[(size_t)ra * PD + c0 + j], bb = P.R[(size_t)rb * PD + c0 + j];
            const float2 c = P.Q[(size_t)ra * PD + c0 + j], d = P.Q[(size_t)rb * PD + c0 + j];
            MR[0][j] = mkv2(a.x, a.y); MR[1][j] = mkv2(bb.x, bb.y);
            MQ[0][j] = mkv2(c.x, c.y); MQ[1][j] = mkv2(d.x, d.y);
        }
    }
    const int own_f = vec_float_index<PD>(row, comp, clip);
    const int rd4 = q * VSL;
    float* stash = reinterpret_cast<float*>(P.stash);
    const size_t pos = (size_t)w * 64 + lane;
    for (int cp = 0; cp < npairs; ++cp) {                         // ut_0 = the columns phi_a (trace 1: "inv_{-1}" = 1)
        const float2 p0 = P.phi0[(size_t)(2 * cp + clip) * PD + row];
        reinterpret_cast<float*>(uvec + (size_t)cp * VEC4)[own_f] = im_lane ? p0.y : p0.x;
    }
    float sv = 0.f, nbuf = 0.f;
    float2 rho_next = P.rho[row];
    __syncthreads();
    for (int k = 0; k <= N; ++k) {
        const int p = k & 1;
        const bool chain = k < N;
        if (chain && (k & (WCH - 1)) == 0) {                      // increments of the next 64 steps, one per lane (model.py:135, 175)
            const int idx = k + lane;
            sv = ((idx + 1 < T ? xr[idx + 1] : 0.f) - (idx < T ? xr[idx] : 0.f)) / A;
        }
        const float2 rho_k = rho_next;
        if (k + 1 < N) rho_next = P.rho[(size_t)(k + 1) * PD + row];
        float n = 1.f;                                            // tr rho'_{k-1} (published before the barrier of the previous iteration)
        if (k >= 1) {
            n = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) n += nrm[p * NW + ww];
        }
        const float inv = k >= 1 ? rsq_newton(fmaxf(n, 1e-12f)) : 1.f;       // model.py:198-203 (sqrt of the trace: columns, not the matrix)
        if (chain) {
            const float sk = wrdl(sv, k & (WCH - 1));
            const v2f s2 = mkv2(sk, sk);
            v2f MM[2][KC];                                        // M_k = Q + s_k R: once per step, shared by every column of the clip
#pragma unroll
            for (int j = 0; j < KC; ++j) {
                MM[0][j] = __builtin_elementwise_fma(MR[0][j], s2, MQ[0][j]);
                MM[1][j] = __builtin_elementwise_fma(MR[1][j], s2, MQ[1][j]);
            }
            float nacc = 0.f;
            for (int cp = 0; cp < npairs; ++cp) {
                const v4f* uv = uvec + ((size_t)p * npairs + cp) * VEC4;
                v2f cRe0 = mkv2(0.f, 0.f), cIm0 = cRe0, cRe1 = cRe0, cIm1 = cRe0;
#pragma unroll
                for (int j = 0; j < KC; ++j) {
                    const v4f x = uv[rd4 + j];
                    col_single(cRe0, cIm0, cRe1, cIm1, MM[0][j], MM[1][j], lo_of(x), hi_of(x));
                }
                const float ut = reinterpret_cast<const float*>(uv)[own_f];
                const float acc = reduce_slices(cRe0, cIm0, cRe1, cIm1, clip1);
                const float y = inv * (ut + acc);
                nacc = fmaf(y, y, nacc);
                stash[wide_stash_vec<PD>((size_t)b * npairs + cp, N, k, 0) + pos] = y;
                const float py = partner16(y, im_lane);
                reinterpret_cast<float*>(uvec + ((size_t)(p ^ 1) * npairs + cp) * VEC4)[own_f] =
                    rho_k.x * y + (im_lane ? rho_k.y : -rho_k.y) * py;       // ut_{k+1} = rho_k y_k (un-normalised)
            }
            float nn = clip_wave_sum(nacc);
            nn += wdpp<0x128>(nn);                                // + the columns of the other parity (lane ^ 8)
            if (lane == 0) nrm[(p ^ 1) * NW + w] = nn;
        }
        if (w == 0 && k >= 1) {                                   // tr rho'_{k-1} rows of the per-clip scalar stash
            const int kn = k - 1;
            if (lane == (kn & (WCH - 1))) nbuf = n;
            if ((kn & (WCH - 1)) == WCH - 1 || kn == N - 1) {
                const int c = kn / WCH;
                if (c * WCH + lane < N) rscal[((size_t)b * NC + c) * 128 + lane] = nbuf;
            }
        }
        wide_barrier();
    }
}

// audio [B][T] -> [B vrank][T]: every column of a clip reads the clip's samples
__global__ void k_rho_expand_audio(const float* __restrict__ audio, float* __restrict__ vaudio, int T, int vrank) {
    const float* src = audio + (size_t)(blockIdx.x / vrank) * T;
    float* dst = vaudio + (size_t)blockIdx.x * T;
    for (int t = threadIdx.x; t < T; t += blockDim.x) dst[t] = src[t];
}
// the clip's |y|^2 rows -> every virtual clip's rows (k_hy_wide scales its fp16 pieces from them: |y_a|^2 <= tr rho')
__global__ void k_rho_spread_n(const float* __restrict__ rscal, float* __restrict__ vscal, int NC, int vrank) {
    const size_t v = blockIdx.x, c = blockIdx.y;
    vscal[(v * NC + c) * 128 + threadIdx.x] = rscal[((v / vrank) * NC + c) * 128 + threadIdx.x];
}
// e = sum over the clip's columns of y_a^dagger H y_a (fixed order), to the clip's row and back to every column's row
__global__ void k_rho_merge_e(float* __restrict__ rscal, float* __restrict__ vscal, int NC, int vrank) {
    const size_t b = blockIdx.x, c = blockIdx.y;
    float e = 0.f;
    for (int a = 0; a < vrank; ++a) e += vscal[((b * vrank + a) * NC + c) * 128 + 64 + threadIdx.x];
    rscal[(b * NC + c) * 128 + 64 + threadIdx.x] = e;
    for (int a = 0; a < vrank; ++a) vscal[((b * vrank + a) * NC + c) * 128 + 64 + threadIdx.x] = e;
}
// zero-padded copy of the columns: [rank][DP] -> [vrank][DP]
__global__ void k_rho_pad_phi(const float2* __restrict__ phi0, float2* __restrict__ vphi, int rank, int vrank, int DP) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < vrank * DP) vphi[idx] = idx < rank * DP ? phi0[idx] : make_float2(0.f, 0.f);
}
// cotangents of the columns: dphi_a = sum over clips of the reverse chain's final g of virtual clip (b, a)
__global__ void k_rho_phi_reduce(const float* __restrict__ gphi, int B, int vrank, int rank, int D, int DP, float* __restrict__ out_re,
                                 float* __restrict__ out_im) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rank * D) return;
    const int a = idx / D, d = idx % D;
    double sr = 0.0, si = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* g = gphi + ((size_t)b * vrank + a) * 2 * DP;
        sr += (double)g[d];
        si += (double)g[DP + d];
    }
    out_re[idx] = (float)sr;
    out_im[idx] = (float)si;
}

// ------------------------------------------------------------------------------------------------------------------------
// PsiCMPS.sample (model.py:242-251, 284-291) for 32 < D <= 128 (round 4: a mode of the wide chain; before, the block sampler re-read
// both matrices from L2 every step).  One workgroup per PAIR of paths, the forward chain's lane layout and broadcast; the step is
//   e = 2 inv^2 Re(ut^dagger R ut)  ->  inc = e dt + noise_k,  samp += inc,  s = inc / A  ->  y_k = inv (ut + Q ut + s R ut),
// with ut = rho_{k-1} y_{k-1} carried un-normalised as in the forward (inv = 1 / |y_{k-1}|, its partial sums ride with the broadcast).
// R ut and Q ut stay separate (the increment that scales R ut depends on <R>), and the expectation needs one more exchange across
// the waves: two LDS-only barriers per step instead of one.
// ------------------------------------------------------------------------------------------------------------------------
template <int PD>
__global__ __launch_bounds__(4 * PD) void k_sample_wide(Dev P, const float* __restrict__ noise, int n_paths, int length,
                                                        float* __restrict__ out) {
    using G = WideGeom<PD>;
    constexpr int NW = G::NW, KC = G::KC, VSL = G::VSL, VEC4 = G::VEC4;
    __shared__ __attribute__((aligned(16))) v4f uvec[2 * VEC4];
    __shared__ __attribute__((aligned(8))) float nrm[2 * NW * 2];
    __shared__ __attribute__((aligned(8))) float ee[NW * 2];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = lane >> 3, i = lane & 7;
    const int rowsel = q >> 2, comp = (q >> 1) & 1, clip = q & 1;
    const bool clip1 = clip != 0, im_lane = comp != 0;
    const int row = 16 * w + 8 * rowsel + i;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < n_paths) ? b0 + 1 : b0;   // an odd count repeats its last path (not stored)
    const bool two = b1 != b0;
    const float* nr0 = noise + (size_t)b0 * length;
    const float* nr1 = noise + (size_t)b1 * length;
    float* orow = out + (size_t)(clip1 ? b1 : b0) * length;
    const bool writer = w == 0 && i == 0 && q < 2 && (!clip1 || two);
    const float A = dev_A(P), dt = P.dt;

    v2f MR[2][KC], MQ[2][KC];
    {
        const int ra = 16 * w + i, rb = ra + 8, c0 = q * KC;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const float2 a = P.R[(size_t)ra * PD + c0 + j], b = P.R[(size_t)rb * PD + c0 + j];
            const float2 c = P.Q[(size_t)ra * PD + c0 + j], d = P.Q[(size_t)rb * PD + c0 + j];
            MR[0][j] = mkv2(a.x, a.y); MR[1][j] = mkv2(b.x, b.y);
            MQ[0][j] = mkv2(c.x, c.y); MQ[1][j] = mkv2(d.x, d.y);
        }
    }
    const int own_f = vec_float_index<PD>(row, comp, clip);
    const int rd4 = q * VSL;
    const float2 p0 = P.psi0[row];
    float ut = im_lane ? p0.y : p0.x;
    reinterpret_cast<float*>(uvec)[own_f] = ut;
    float nz0 = 0.f, nz1 = 0.f;                                   // noise of the current 64 steps, lane <-> step
    float samp = 0.f;                                             // model.py:244 batch_zeros (this lane's path)
    float2 rho_next = P.rho[row];
    __syncthreads();

    for (int k = 0; k < length; ++k) {
        const int p = k & 1, kl = k & (WCH - 1);
        if (kl == 0) {
            const int idx = k + lane;
            nz0 = idx < length ? nr0[idx] : 0.f;
            nz1 = idx < length ? nr1[idx] : 0.f;
        }
        const float2 rho_k = rho_next;
        if (k + 1 < P.N) rho_next = P.rho[(size_t)(k + 1) * PD + row];
        float n0 = 1.f, n1 = 1.f;
        if (k >= 1) {
            n0 = n1 = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) {
                const float2 t = *reinterpret_cast<const float2*>(&nrm[(p * NW + ww) * 2]);
                n0 += t.x; n1 += t.y;
            }
        }
        const float inv = k >= 1 ? rsq_newton(fmaxf(clip1 ? n1 : n0, 1e-12f)) : 1.f;      // model.py:289 of the step before
        v2f rRe0 = mkv2(0.f, 0.f), rIm0 = rRe0, rRe1 = rRe0, rIm1 = rRe0;
        v2f qRe0 = rRe0, qIm0 = rRe0, qRe1 = rRe0, qIm1 = rRe0;
        const v4f* uv = uvec + p * VEC4 + rd4;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const v4f x = uv[j];
            col_two(rRe0, rIm0, rRe1, rIm1, qRe0, qIm0, qRe1, qIm1, MR[0][j], MQ[0][j], MR[1][j], MQ[1][j], lo_of(x), hi_of(x));
        }
        const float vs = reduce_slices(rRe0, rIm0, rRe1, rIm1, clip1);      // (R ut), (Q ut): this lane's (row, component, path)
        const float qs = reduce_slices(qRe0, qIm0, qRe1, qIm1, clip1);
        const float ep = clip_wave_sum(ut * vs);
        if (i == 0 && q < 2) ee[w * 2 + clip] = ep;
        wide_barrier();
        float e0 = 0.f, e1 = 0.f;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const float2 t = *reinterpret_cast<const float2*>(&ee[ww * 2]);
            e0 += t.x; e1 += t.y;
        }
        const float e = 2.0f * ((clip1 ? e1 : e0) * inv) * inv;   // _expectation on the normalised state (model.py:319-325)
        const float inc = e * dt + (clip1 ? wrdl(nz1, kl) : wrdl(nz0, kl));   // model.py:286
        samp += inc;                                              // :287
        const float s = inc / A;                                  // :288 -> :303
        const float y = inv * (ut + (qs + s * vs));
        const float nn = clip_wave_sum(y * y);
        if (i == 0 && q < 2) nrm[((p ^ 1) * NW + w) * 2 + clip] = nn;
        const float py = partner16(y, im_lane);
        ut = rho_k.x * y + (im_lane ? rho_k.y : -rho_k.y) * py;   // rho_k y_k, normalised in the next step
        reinterpret_cast<float*>(uvec + (p ^ 1) * VEC4)[own_f] = ut;
        if (writer) orow[k] = A * samp;                           // model.py:251
        wide_barrier();
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// H y_k, e_k = y_k . (H y_k) for every (clip, step) at once (training forward, after the chain kernel has stashed y_k):
//   Re(H y) = [H_re | -H_im] [y_re; y_im],   Im(H y) = [H_re | -H_im] [y_im; -y_re]
// as v_mfma_f32_32x32x16_bf16 GEMMs, A = the real form of H (32 rows per wave, K = 2 PD, split EXACTLY into three bf16 pieces once
// per workgroup and kept in registers), B = 32 columns (step of an 8-step unit, clip, {Re, Im} form) split the same way on the
// fly; the six significant piece products give an fp32-faithful H y (24 operand bits, fp32 accumulate).  One workgroup per
// (pair, 512-step chunk): nothing here is serial in time.  Writes the H y half of the stash (lane order) and e_k into the
// scalar stash; k_loss_wide then accumulates -log(1 + e x / A) sequentially in float32 (model.py:279, 294).
// ------------------------------------------------------------------------------------------------------------------------
namespace {

constexpr int HU = 8;                      // steps per unit (32 MFMA columns)
constexpr int HCHUNK = 512;                // steps per workgroup

template <int PD>
struct HyGeom {
    static constexpr int PROW = 2 * PD + 16;                      // bytes of one (step, clip, component) row of bf16 + bank-spread padding
    static constexpr int PIECE = HU * 4 * PROW;                   // bytes of one piece of one unit: [step][clip][comp] rows
    static constexpr int FROW = PD + 4;                           // floats of one float32 row + padding
    static constexpr int FBUF = HU * 4 * FROW;                    // floats of one float32 unit buffer
    static constexpr size_t LDS = (size_t)2 * 3 * PIECE + (size_t)5 * FBUF * 4 + 2 * (PD / 32) * 16 * 4;   // pieces x2, y x3, H y x2, e
};                                                                // (the fp16 form has two pieces per buffer and leaves the third unused)
// Where row (step j, clip, component) = 4 j + c, c = 2 clip + comp, of a unit sits in its LDS buffers.  Consecutive slots are 4 banks
// apart (PROW, FROW = 4 mod 64 dwords).  That suits the MFMA-side ds_read_b128 accesses (a 16-lane group = 16 rows x 16 B fills the 64
// banks) but not the build / write-out side, where a lane group touches the FOUR rows of one step with 8 - 16 consecutive dwords each and
// stores bank modulo 32: rows 4 banks apart overlap 2- to 4-way (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE 0.30,
// profiles/r5_c5wide_pmc_summary.json).  slot = 8 (j >> 1) + 2 c + (j & 1): the rows of a step sit two slots = 8 banks apart (stores
// of 8 dwords per row conflict-free, the 16-dword H y reads 2-way), and the row sets of the ds_read_b128 lane groups ({0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}: MI355X_MICROARCH.md, LDS) still map to sixteen different slots modulo 16.
#ifdef CMPS_DIAG_HY_NO_SLOT
__device__ __forceinline__ constexpr int hy_slot(int row) { return row; }
#else
__device__ __forceinline__ constexpr int hy_slot(int row) { return 8 * (row >> 3) + 2 * (row & 3) + ((row >> 2) & 1); }
#endif

}  // namespace

// Five units are in flight per workgroup, one barrier per unit:  rows of unit u + 2 requested | bf16 pieces and float32 rows of unit
// u + 1 built | MFMAs of unit u | e partials and H y rows of unit u - 1 from the accumulators (two accumulator pairs, by unit parity)
// | H y rows and e of unit u - 2 out to memory.  The issue order is written out as in k_grad_gemm (cmps_grad_gemm.h): slot t = MFMA t
// + at most one small slice of the other four stages, fenced.  The sign of the Im form's second K half (-y_re) is applied to the
// accumulator of that half (out = acc0 +- acc1 per column form), not to the operand reads.
// F16 (CMPS_RANK1_F16X2 / DEFAULT): two fp16 pieces per operand instead of three bf16 ones and three products instead of six, as in
// k_grad_gemm (cmps_grad_gemm.h): H is scaled by a power of two from its largest entry, the y rows of the workgroup's 512 steps from
// their largest |y|^2 (the chain kernel's scalar stash); the product is unscaled where it leaves the accumulators.
template <int PD, bool F16>
__global__ __launch_bounds__(2 * PD, 1) void k_hy_wide(Dev P) {
    using HG = HyGeom<PD>;
    using gg::static_for;
    constexpr int PWV = PD / 32, KT = 2 * PD / 16, PROW = HG::PROW, PIECE = HG::PIECE, FROW = HG::FROW, FBUF = HG::FBUF;
    constexpr int NPR = F16 ? 3 : 6;                              // piece products
    constexpr int NM = NPR * KT;                                  // MFMAs per unit
    extern __shared__ __attribute__((aligned(16))) unsigned char wide_lds[];
    unsigned char* pcs = wide_lds;                                          // [2 buffers][3 pieces][PIECE]
    float* yf = reinterpret_cast<float*>(wide_lds + 2 * 3 * PIECE);         // [3][FBUF]: y, float32
    float* hf = yf + 3 * FBUF;                                              // [2][FBUF]: H y, float32
    float* eacc = hf + 2 * FBUF;                                            // [2][PWV][16]
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int N = P.N, NC = (N + WCH - 1) / WCH;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;
    const bool two = b1 != b0;
    const int k_lo = blockIdx.y * HCHUNK, k_hi = (k_lo + HCHUNK < N) ? k_lo + HCHUNK : N;
    const int NU = (k_hi - k_lo + HU - 1) / HU;
    float* stash = reinterpret_cast<float*>(P.stash);
    const int mr = lane & 31, mh = lane >> 5;
    // this lane's column of a unit: (step, clip, form)
    const int cs = mr >> 2, cc = (mr >> 1) & 1, cf = mr & 1;
    const float sgn = cf ? -1.f : 1.f;

    // ---- A operand: row 32 w + mr of [H_re | -H_im], K values 16 t + 8 mh .. + 7, three bf16 pieces (two fp16 pieces) ----
    bf8w Ah[KT], Am[F16 ? 1 : KT], Al[KT];
    float sH = 1.f, sB = 1.f;
    if constexpr (F16) {
        const int row = 32 * w + mr;
        float mH = 0.f, mN = 1.f;
        for (int e = 0; e < PD / 2; ++e) {                        // this lane's half of its row of H
            const int j = mh * (PD / 2) + e;
            const float2 r = P.R[(size_t)row * PD + j], rt = P.RT[(size_t)row * PD + j];
            mH = fmaxf(mH, fmaxf(fabsf(r.x + rt.x), fabsf(r.y - rt.y)));
        }
        for (int e = tid; e < 2 * (k_hi - k_lo); e += 2 * PD) {   // |y_k|^2 of the workgroup's steps (written by k_fwd_wide)
            const int k = k_lo + (e >> 1);
            mN = fmaxf(mN, P.scal[((size_t)((e & 1) ? b1 : b0) * NC + k / WCH) * 128 + (k & (WCH - 1))]);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            mH = fmaxf(mH, __shfl_xor(mH, off, 64));
            mN = fmaxf(mN, __shfl_xor(mN, off, 64));
        }
        if (lane == 0) { eacc[2 * w] = mH; eacc[2 * w + 1] = mN; }
        __syncthreads();
#pragma unroll
        for (int ww = 0; ww < PWV; ++ww) { mH = fmaxf(mH, eacc[2 * ww]); mN = fmaxf(mN, eacc[2 * ww + 1]); }
        __syncthreads();
        auto uni = [](float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
        sH = uni(gg::pow2_scale(mH, 15));
        sB = uni(gg::pow2_scale(sqrtf(mN), 13));
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            unsigned ph[4], pl[4];
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                float v[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int k = 16 * t + 8 * mh + e + i, part = k / PD, j = k % PD;
                    const float2 r = P.R[(size_t)row * PD + j], rt = P.RT[(size_t)row * PD + j];
                    v[i] = (part == 0 ? r.x + rt.x : -(r.y - rt.y)) * sH;
                }
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                ph[e >> 1] = gg::cvt_pk_f16(v[0], v[1]);
                const h2 hh = __builtin_bit_cast(h2, ph[e >> 1]);
                pl[e >> 1] = gg::cvt_pk_f16(v[0] - (float)hh.x, v[1] - (float)hh.y);
            }
            Ah[t] = __builtin_bit_cast(bf8w, u4w{ph[0], ph[1], ph[2], ph[3]});
            Al[t] = __builtin_bit_cast(bf8w, u4w{pl[0], pl[1], pl[2], pl[3]});
        }
    } else {
        const int row = 32 * w + mr;
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            unsigned ph[4], pm[4], pl[4];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * t + 8 * mh + e, part = k / PD, j = k % PD;
                const float2 r = P.R[(size_t)row * PD + j], rt = P.RT[(size_t)row * PD + j];    // RT[i][j] = R[j][i]
                const float v = part == 0 ? r.x + rt.x : -(r.y - rt.y);                          // H = R + R^dagger
                unsigned h, m, l;
                split3(v, h, m, l);
                if (e & 1) {
                    ph[e >> 1] = __builtin_amdgcn_perm(h, ph[e >> 1], 0x07060302u);
                    pm[e >> 1] = __builtin_amdgcn_perm(m, pm[e >> 1], 0x07060302u);
                    pl[e >> 1] = __builtin_amdgcn_perm(l, pl[e >> 1], 0x07060302u);
                } else {
                    ph[e >> 1] = h; pm[e >> 1] = m; pl[e >> 1] = l;
                }
            }
            Ah[t] = __builtin_bit_cast(bf8w, u4w{ph[0], ph[1], ph[2], ph[3]});
            Am[t] = __builtin_bit_cast(bf8w, u4w{pm[0], pm[1], pm[2], pm[3]});
            Al[t] = __builtin_bit_cast(bf8w, u4w{pl[0], pl[1], pl[2], pl[3]});
        }
    }
    // ---- build / write-out role: positions 2 tid and 2 tid + 1 of a stash vector: two adjacent rows of one (component, clip) ----
    const int ppos = 2 * tid, pl_ = ppos & 63;
    const int pq = pl_ >> 3, pi = pl_ & 7;
    const int pcomp = (pq >> 1) & 1, pclip = pq & 1;
    const int prow = 16 * (ppos >> 6) + 8 * (pq >> 2) + pi;      // even; the second position is row prow + 1
    // buffer descriptors (SGPR row offset + a loop-invariant lane offset; a store to be dropped gets a lane offset beyond the
    // descriptor's range instead of a branch): the pair's stash rows, and the scalar stash
    float* pair_rows = stash + wide_stash_vec<PD>(blockIdx.x, N, 0, 0);
    const auto rs_st = __builtin_amdgcn_make_buffer_rsrc(pair_rows, 0, N * (8 * PD * 4), 0x00020000);
    const auto rs_sc = __builtin_amdgcn_make_buffer_rsrc(P.scal, 0, (int)((size_t)P.B * NC * 128 * 4), 0x00020000);
    const int voff_y = ppos * 4, voff_h = (4 * PD + ppos) * 4;

    // rows of a unit (steps k_lo + 8 u ..).  Unclamped loads: rows a few steps above the pair's range lie inside the caller's
    // workspace, and every value derived from them is discarded by a select.
    float2 Y[HU];
    auto fetch_row = [&](int u, int j) {                          // (rows past the pair's last one read as zero: the descriptor's range)
        Y[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_st, voff_y, (k_lo + HU * u + j) * (8 * PD * 4), 0));
    };
    // step j of unit u -> bf16 pieces (buffer u & 1) + float32 row (buffer u % 3); three parts
    unsigned sh[3];
    float sr0 = 0.f, sr1 = 0.f, sv0 = 0.f, sv1 = 0.f;
    auto prep_part = [&](int u, int u3, int j, int part) {
        if (part == 0) {
            const bool in = k_lo + HU * u + j < k_hi;
            sv0 = in ? Y[j].x : 0.f; sv1 = in ? Y[j].y : 0.f;
            if constexpr (F16) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                const float t0 = sv0 * sB, t1 = sv1 * sB;
                sh[0] = gg::cvt_pk_f16(t0, t1);
                const h2 hh = __builtin_bit_cast(h2, sh[0]);
                sr0 = t0 - (float)hh.x;
                sr1 = t1 - (float)hh.y;
            } else {
                sr0 = sv0 - __uint_as_float(__float_as_uint(sv0) & 0xFFFF0000u);
                sr1 = sv1 - __uint_as_float(__float_as_uint(sv1) & 0xFFFF0000u);
            }
        } else if (part == 1 && F16) {
            sh[1] = gg::cvt_pk_f16(sr0, sr1);
        } else if (part == 1) {
            const float q0 = sr0 - __uint_as_float(__float_as_uint(sr0) & 0xFFFF0000u);
            const float q1 = sr1 - __uint_as_float(__float_as_uint(sr1) & 0xFFFF0000u);
            sh[0] = pack_hi16(__float_as_uint(sv0), __float_as_uint(sv1));
            sh[1] = pack_hi16(__float_as_uint(sr0), __float_as_uint(sr1));
            sh[2] = pack_hi16(__float_as_uint(q0), __float_as_uint(q1));
        } else {
            const int rowi = hy_slot((j * 2 + pclip) * 2 + pcomp);
            unsigned char* d = pcs + (size_t)(u & 1) * 3 * PIECE + rowi * PROW + prow * 2;     // rows prow, prow + 1: one dword per piece
            *reinterpret_cast<unsigned*>(d) = sh[0];
            *reinterpret_cast<unsigned*>(d + PIECE) = sh[1];
            if constexpr (!F16) *reinterpret_cast<unsigned*>(d + 2 * PIECE) = sh[2];
            *reinterpret_cast<float2*>(&yf[(size_t)u3 * FBUF + rowi * FROW + prow]) = make_float2(sv0, sv1);
        }
    };

    // one unit.  PAR = u & 1 (static: accumulator pair and LDS buffers of the stages), u3 = u % 3.
    f16w acc0[2], acc1[2];                                         // [unit parity]: K half 0 (H_re), K half 1 (-H_im)
    float ep_run = 0.f;
    float4 hv = make_float4(0.f, 0.f, 0.f, 0.f);
    const float usc = (1.0f / sH) * (1.0f / sB), usgn = sgn * usc;   // exact powers of two (1 without the fp16 scales)
    auto run_unit = [&](auto par_c, auto mac_c, int u, int u3) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool MAC = decltype(mac_c)::value;
        constexpr int S_W0 = 0, S_E0 = S_W0 + 9, S_P0 = S_E0 + 9, S_F0 = S_P0 + 24, NS = S_F0 + 8;
        const unsigned char* pb = pcs + (size_t)PAR * 3 * PIECE;
        const int up3 = u3 == 0 ? 2 : u3 - 1, un3 = u3 == 2 ? 0 : u3 + 1;    // (u - 1) % 3, (u + 1) % 3
        const int rowi = hy_slot((cs * 2 + cc) * 2 + cf);
        const float* yr = yf + (size_t)up3 * FBUF + rowi * FROW + 32 * w + 4 * mh;
        float* hr = hf + (size_t)(1 - PAR) * FBUF + rowi * FROW + 32 * w + 4 * mh;
        const int kw = k_lo + HU * (u - 2);                        // first step of the unit being written out
        const float* hb = hf + (size_t)PAR * FBUF;
        bf8w Bq[2][3];
        auto read_b = [&](int t, int buf) {
            const int part = (16 * t) / PD, j0 = (16 * t) % PD + 8 * mh;
            const int comp = part == 0 ? cf : 1 - cf;              // Re form: [y_re; y_im]; Im form: [y_im; y_re] (sign: see the epilogue)
            const unsigned char* src = pb + hy_slot((cs * 2 + cc) * 2 + comp) * PROW + j0 * 2;
            Bq[buf][0] = __builtin_bit_cast(bf8w, *reinterpret_cast<const u4w*>(src));
            Bq[buf][1] = __builtin_bit_cast(bf8w, *reinterpret_cast<const u4w*>(src + PIECE));
            if constexpr (!F16) Bq[buf][2] = __builtin_bit_cast(bf8w, *reinterpret_cast<const u4w*>(src + 2 * PIECE));
        };
        auto slice = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I < S_E0) {                             // unit u - 2: H y rows and e out (lane order: coalesced)
                if constexpr (I < 8) {
                    constexpr int j = I;
                    const float2 v = *reinterpret_cast<const float2*>(&hb[hy_slot((j * 2 + pclip) * 2 + pcomp) * FROW + prow]);
                    const bool ok = u >= 2 && kw + j < k_hi;
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2w, v), rs_st, ok ? voff_h : 0x7fffffff,
                                                          ok ? (kw + j) * (8 * PD * 4) : 0, 0);
                } else {
                    const int j = (tid >> 1) & 7, cl = tid & 1, k = kw + j;
                    float e = 0.f;
#pragma unroll
                    for (int ww = 0; ww < PWV; ++ww) e += eacc[(PAR * PWV + ww) * 16 + (tid & 15)];
                    const bool ok = tid < 16 && u >= 2 && k < k_hi && (cl == 0 || two);
                    const int so = (int)((((size_t)(cl ? b1 : b0) * NC + (ok ? k / WCH : 0)) * 128 + 64 + (k & (WCH - 1))) * 4);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, e), rs_sc, ok ? so : 0x7fffffff, 0, 0);
                }
            } else if constexpr (I < S_P0) {                      // unit u - 1: H y = acc0 +- acc1, e partial, float32 rows into hf
                constexpr int x = I - S_E0;                       // C/D layout: column = lane & 31, rows (r & 3) + 8 (r >> 2) + 4 mh
                if constexpr (x < 8) {
                    constexpr int g = x / 2;
                    if constexpr ((x & 1) == 0) {
                        if constexpr (F16)
                            hv = make_float4(fmaf(acc0[1 - PAR][4 * g], usc, usgn * acc1[1 - PAR][4 * g]), fmaf(acc0[1 - PAR][4 * g + 1], usc, usgn * acc1[1 - PAR][4 * g + 1]),
                                             fmaf(acc0[1 - PAR][4 * g + 2], usc, usgn * acc1[1 - PAR][4 * g + 2]), fmaf(acc0[1 - PAR][4 * g + 3], usc, usgn * acc1[1 - PAR][4 * g + 3]));
                        else
                            hv = make_float4(acc0[1 - PAR][4 * g] + sgn * acc1[1 - PAR][4 * g], acc0[1 - PAR][4 * g + 1] + sgn * acc1[1 - PAR][4 * g + 1],
                                             acc0[1 - PAR][4 * g + 2] + sgn * acc1[1 - PAR][4 * g + 2], acc0[1 - PAR][4 * g + 3] + sgn * acc1[1 - PAR][4 * g + 3]);
                        if constexpr (g == 0) ep_run = 0.f;
                    } else {
                        const v4f yv = *reinterpret_cast<const v4f*>(yr + 8 * g);
                        ep_run += yv.x * hv.x + yv.y * hv.y + yv.z * hv.z + yv.w * hv.w;
                        *reinterpret_cast<v4f*>(hr + 8 * g) = v4f{hv.x, hv.y, hv.z, hv.w};
                    }
                } else {
                    float ep = swap32_add(ep_run, ep_run);        // + the other row half
                    ep += wdpp<0xB1>(ep);                         // + the other component (column ^ 1)
                    if (mh == 0 && cf == 0) eacc[((1 - PAR) * PWV + w) * 16 + (mr >> 1)] = ep;
                }
            } else if constexpr (I < S_F0) {                      // unit u + 1: pieces and float32 rows from the fetched rows
                constexpr int x = I - S_P0;
                prep_part(u + 1, un3, x / 3, x % 3);
            } else {                                              // unit u + 2: request its rows
                fetch_row(u + 2, I - S_F0);
            }
        };
        if constexpr (MAC) {
            read_b(0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_for<0, (MAC ? NM : NS)>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (MAC) {
                constexpr int kt = t / NPR, pr = t % NPR, buf = kt & 1;
                if constexpr (pr == 0 && kt + 1 < KT) read_b(kt + 1, 1 - buf);
                // piece products a + b <= 2: lo hi', hi lo', mid mid', mid hi', hi mid', hi hi'  (fp16: lo hi', hi lo', hi hi')
                const bf8w av = pr == 0 ? Al[kt] : (!F16 && (pr == 2 || pr == 3)) ? Am[F16 ? 0 : kt] : Ah[kt];
                const bf8w bv = F16 ? (pr == 1 ? Bq[buf][1] : Bq[buf][0])
                                    : (pr == 1 ? Bq[buf][2] : (pr == 2 || pr == 4) ? Bq[buf][1] : Bq[buf][0]);
                constexpr f16w zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if constexpr (kt < KT / 2) acc0[PAR] = gg::mma<F16>(av, bv, t == 0 ? zero : acc0[PAR]);
                else acc1[PAR] = gg::mma<F16>(av, bv, t == NPR * (KT / 2) ? zero : acc1[PAR]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<(t * NS) / NM, ((t + 1) * NS) / NM>(slice);
            } else {
                slice(tc);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

#pragma unroll
    for (int j = 0; j < HU; ++j) fetch_row(0, j);
#pragma unroll
    for (int x = 0; x < 24; ++x) prep_part(0, 0, x / 3, x % 3);
#pragma unroll
    for (int j = 0; j < HU; ++j) fetch_row(1, j);
    __syncthreads();
    // units 0 .. NU + 1: the last two only drain the pipeline (their MFMAs run on zero columns)
    int u3 = 0;
    for (int u = 0; u < NU + 2; u += 2) {
        run_unit(std::integral_constant<int, 0>{}, std::true_type{}, u, u3);
        u3 = u3 == 2 ? 0 : u3 + 1;
        __syncthreads();
        run_unit(std::integral_constant<int, 1>{}, std::true_type{}, u + 1, u3);
        u3 = u3 == 2 ? 0 : u3 + 1;
        __syncthreads();
    }
}

// loss_b = sum_k -log(1 + (e_k x_k) / A), accumulated sequentially in float32 in time order (model.py:279, 294): one wavefront per
// clip, the logarithms of 64 steps at a time (one step per lane), then 64 ordered adds
template <bool LEGACY>
__global__ __launch_bounds__(64) void k_loss_wide(Dev P, const float* __restrict__ audio, float* __restrict__ loss_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int N = P.N, NC = (N + WCH - 1) / WCH;
    const float* xr = audio + (size_t)b * P.T;
    float* sc = P.scal + (size_t)b * NC * 128;
    const float A = dev_A(P);
    float loss = 0.f;
    float en = lane < N ? sc[64 + lane] : 0.f, x0n = lane < N ? xr[lane] : 0.f, x1n = lane < N ? xr[lane + 1] : 0.f;
    float nn = (LEGACY && lane < N) ? sc[lane] : 1.f;
    float f_below = LEGACY ? 2.0f * P.R[0].x : 0.f, n_below = 1.f;     // LEGACY: y^dagger H y and |y|^2 of the step below the chunk (psi_0 = e_0)
    for (int c = 0; c < NC; ++c) {
        float e = en;
        const float inc = x1n - x0n, nv = nn;
        const bool in = c * WCH + lane < N;
        const int idx = (c + 1) * WCH + lane;
        if (c + 1 < NC) {
            en = idx < N ? sc[(size_t)(c + 1) * 128 + 64 + lane] : 0.f;
            if (LEGACY) nn = idx < N ? sc[(size_t)(c + 1) * 128 + lane] : 1.f;
            x0n = idx < N ? xr[idx] : 0.f;
            x1n = idx < N ? xr[idx + 1] : 0.f;
        }
        float lv;
        if constexpr (LEGACY) {
            // e_k = psi_k^dagger (R + R^T) psi_k on the normalised state of the step below (graph.pbtxt:11857-12661), loss += (x_k - e_k)^2 / 2
            // (:12685-12819); the e row is REPLACED by the legacy e_k (what the reverse scan and the gradient GEMM read), as the wave kernels do
            float fb = __shfl_up(e, 1, 64), nb = __shfl_up(nv, 1, 64);
            if (lane == 0) { fb = f_below; nb = n_below; }
            f_below = wrdl(e, WCH - 1);
            n_below = wrdl(nv, WCH - 1);
            const float invb = 1.0f / sqrtf(fmaxf(nb, 1e-12f));       // graph.pbtxt:14350-14594
            e = (fb * invb) * invb;
            if (in) sc[(size_t)c * 128 + 64 + lane] = e;
            const float d = inc - e;
            lv = in ? d * d / 2.0f : 0.f;
        } else {
            lv = in ? -logf(1.0f + (e * inc) / A) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < WCH; ++j) loss += wrdl(lv, j);
    }
    if (lane == 0) loss_out[b] = loss;
}

// ------------------------------------------------------------------------------------------------------------------------
// reverse scan
//   yhat = y_k inv_k;  yhb = conj(rho_k) g;  ybar = (yhb - ok yhat rad_{k+1}) inv_k + te_k H y_k
//   g    = ybar + (Q + s_k R^dagger) ybar
//   fbar += dt_k Im(g conj(u_{k+1}));   Abar = -(sum zbar e x) / A^2 - (sum Re(u_k^dagger (Q + s R^dagger) ybar_k)) / A  (abar_fix)
// rad_{k+1} = Re(u_{k+1}^dagger g_{k+1}) = te_{k+1} e_{k+1} (Euler: everything downstream of u_{k+1} is scale invariant except
// the loss term of step k + 1, homogeneous of degree 2), 0 behind the last step.
// ------------------------------------------------------------------------------------------------------------------------
// LEGACY: Q^dagger instead of Q (the legacy Q is not Hermitian) and the legacy per-step scalars -- s = dt x_k, te_k = 2 (e_k - x_k); the
// loss term of step k + 1 looks at the NORMALISED y_k, so it enters ybar_k with the coefficient te_{k+1} inv_k^2 on H y_k and the radial
// part rad_{k+1} = te_{k+1} e_{k+1} (everything else downstream of psi_{k+1} is scale invariant, as in the PsiCMPS arithmetic).
template <int PD, bool LEGACY = false>
__global__ __launch_bounds__(4 * PD) void k_bwd_wide(Dev P, const float* __restrict__ audio) {
    using G = WideGeom<PD>;
    constexpr int NW = G::NW, KC = G::KC, VSL = G::VSL, VEC4 = G::VEC4;
    __shared__ __attribute__((aligned(16))) v4f vec[2][VEC4];
    __shared__ __attribute__((aligned(16))) v4f tab[NW][2][WCH][2][2];     // [wave][chunk parity][step][clip][half]
    __shared__ float redA[NW];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int q = lane >> 3, i = lane & 7;
    const int rowsel = q >> 2, comp = (q >> 1) & 1, clip = q & 1;
    const bool clip1 = clip != 0, im_lane = comp != 0;
    const int row = 16 * w + 8 * rowsel + i;
    const int N = P.N, T = P.T, NC = (N + WCH - 1) / WCH;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;
    const bool two = b1 != b0;
    const float wq = (!clip1 || two) ? 1.f : 0.f;                 // weight of this lane's clip (0: the repeated clip)
    const float* xr0 = audio + (size_t)b0 * T;
    const float* xr1 = audio + (size_t)b1 * T;
    const float* sc0 = P.scal + ((size_t)b0 * NC) * 128;
    const float* sc1 = P.scal + ((size_t)b1 * NC) * 128;
    const float A = dev_A(P);
    const float sgn = im_lane ? 1.f : -1.f;                       // (rho x)_own = rho_re x_own + sgn rho_im x_partner

    v2f MD[2][KC], MQ[2][KC];                                     // R^dagger and Q (Hermitian)
    {
        const int ra = 16 * w + i, rb = ra + 8, c0 = q * KC;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const float2 a = P.RT[(size_t)ra * PD + c0 + j], b = P.RT[(size_t)rb * PD + c0 + j];   // R^dagger[i][j] = conj(RT[i][j])
            if constexpr (LEGACY) {                               // Q^dagger[i][j] = conj(QT[i][j])
                const float2 c = P.QT[(size_t)ra * PD + c0 + j], d = P.QT[(size_t)rb * PD + c0 + j];
                MQ[0][j] = mkv2(c.x, -c.y); MQ[1][j] = mkv2(d.x, -d.y);
            } else {
                const float2 c = P.Q[(size_t)ra * PD + c0 + j], d = P.Q[(size_t)rb * PD + c0 + j];
                MQ[0][j] = mkv2(c.x, c.y); MQ[1][j] = mkv2(d.x, d.y);
            }
            MD[0][j] = mkv2(a.x, -a.y); MD[1][j] = mkv2(b.x, -b.y);
        }
    }
    const int own_f = vec_float_index<PD>(row, comp, clip);
    const int rd4 = q * VSL;
    const float* stf = reinterpret_cast<const float*>(P.stash);
    float* ybs = reinterpret_cast<float*>(P.gops);
    const size_t pos = (size_t)w * 64 + lane;

    // every wave builds its own copy of a chunk's scalar rows (no cross-wave hand-over): lane <-> step
    float accA = 0.f, svA0 = 0.f, svA1 = 0.f, svB0 = 0.f, svB1 = 0.f;      // s of chunk parity 0 / 1, lane <-> step
    float te_above0 = 0.f, te_above1 = 0.f;                       // LEGACY: te of the first step of the chunk above (no step N: 0)
    auto chunk_rows = [&](int cj) {
        const int idx = cj * WCH + lane;
        const bool in = idx < N;
        const float dtv = in ? P.dtk[idx] : 0.f;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) {
            const float* xr = qq ? xr1 : xr0;
            const float* sc = qq ? sc1 : sc0;
            const float x0 = idx < T ? xr[idx] : 0.f, x1 = idx + 1 < T ? xr[idx + 1] : 0.f;
            const float nv = in ? sc[(size_t)cj * 128 + lane] : 1.f;
            const float ev = in ? sc[(size_t)cj * 128 + 64 + lane] : 0.f;
            StepScal r;
            if constexpr (LEGACY) {
                const float inc = x1 - x0;
                const float tev = in ? 2.0f * (ev - inc) : 0.f;   // te_k = 2 ebar_k, ebar_k = e_k - x_k (the e rows hold the legacy e_k: k_loss_wide)
                float ten = __shfl_down(tev, 1, 64);               // te_{k+1}
                if (lane == 63) ten = qq ? te_above1 : te_above0;
                if (qq) te_above1 = wrdl(tev, 0); else te_above0 = wrdl(tev, 0);
                r.s = P.dt * inc;
                r.inv = rsq_newton(fmaxf(nv, 1e-12f));
                r.ok = nv > 1e-12f ? 1.f : 0.f;
                r.te = ten * r.inv * r.inv;                        // the coefficient of H y_k in ybar_k
                r.rad = tev * ev;
                r.zex = 0.f;
            } else {
                r = step_scalars(x1 - x0, nv, ev, A);
            }
            tab[w][cj & 1][lane][qq][0] = v4f{r.s, r.inv, r.ok, r.te};
            tab[w][cj & 1][lane][qq][1] = v4f{r.rad, dtv, 0.f, 0.f};
            if (cj & 1) { if (qq) svB1 = r.s; else svB0 = r.s; } else { if (qq) svA1 = r.s; else svA0 = r.s; }
            // (RhoCMPS on virtual clips: z = e x / A belongs to the clip, not to its columns -- counted with the clip's first column)
            if (in && w == 0 && (qq == 0 || two) && (!P.phi0 || (2 * blockIdx.x + qq) % P.phi_rank == 0)) accA += r.zex;
        }
    };
    chunk_rows((N - 1) / WCH);
    if ((N - 1) / WCH > 0) chunk_rows((N - 1) / WCH - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    auto tab_row = [&](int k, int half) { return tab[w][(k / WCH) & 1][k & (WCH - 1)][clip][half]; };
    // stash row k for this lane: y own, y of the partner component (lane ^ 16), H y own; clamped, unconditional loads
    struct Row { float y, yp, h; };
    auto row_at = [&](int k) {
        const int kc = k > 0 ? k : 0;
        const float* base = stf + wide_stash_vec<PD>(blockIdx.x, N, kc, 0);
        Row r;
        r.y = base[pos];
        r.yp = base[pos ^ 16];
        r.h = base[4 * PD + pos];
        return r;
    };
    auto rho_row = [&](int k) { return P.rho[(size_t)(k > 0 ? k : 0) * PD + row]; };

    const int k0 = N - 1;
    Row ring0 = row_at(k0 - ((k0 - 0) & 3)), ring1 = row_at(k0 - ((k0 - 1) & 3));
    Row ring2 = row_at(k0 - ((k0 - 2) & 3)), ring3 = row_at(k0 - ((k0 - 3) & 3));
    float2 rh = rho_row(k0), rhp = rho_row(k0 - 1);
    v4f S0 = tab_row(k0, 0), S1 = tab_row(k0, 1);
    v4f SP0 = tab_row(k0 > 0 ? k0 - 1 : 0, 0), SP1 = tab_row(k0 > 0 ? k0 - 1 : 0, 1);
    float g = 0.f, unext = 0.f, facc = 0.f, accS = 0.f;
    float ymx = 0.f;                                              // max |ybar| (the gradient GEMM's fp16 operand scale)
    float c3;
    {
        const Row cur = row_at(k0);
        c3 = S0.w * cur.h;                                        // rad_N = 0
    }
    const float2 ps = P.phi0 ? P.phi0[(size_t)((2 * blockIdx.x + clip) % P.phi_rank) * PD + row] : P.psi0[row];
    const float ps0 = im_lane ? ps.y : ps.x;
    int p = 0;
    __syncthreads();

    auto step = [&](int k, Row& CUR, const Row& PRV) {
        const int km2 = k > 1 ? k - 2 : 0;
        if ((k & (WCH - 1)) == WCH - 1 && k != N - 1 && k >= WCH) chunk_rows(k / WCH - 1);   // entering chunk k / WCH: build the one below
        const float2 nrh = rho_row(k - 2);
        // ---- the chain ----
        const float pg = partner16(g, im_lane);
        const float hb = rh.x * g - sgn * rh.y * pg;              // conj(rho_k) g
        const float yb = fmaf(hb, S0.y, c3);
        reinterpret_cast<float*>(vec[p])[own_f] = yb;
        ybs[wide_ybar_vec<PD>(blockIdx.x, N, k) + pos] = yb;
        ymx = fmaxf(ymx, fabsf(yb));
        wide_barrier();
        const int kl = k & (WCH - 1);
        const bool par = ((k / WCH) & 1) != 0;
        const v2f s2 = mkv2(wrdl(par ? svB0 : svA0, kl), wrdl(par ? svB1 : svA1, kl));
        v2f dRe0 = mkv2(0.f, 0.f), dIm0 = dRe0, dRe1 = dRe0, dIm1 = dRe0;
        const v4f* yv = vec[p] + rd4;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const v4f x = yv[j];
            col_merged(dRe0, dIm0, dRe1, dIm1, MD[0][j], MQ[0][j], MD[1][j], MQ[1][j], s2, lo_of(x), hi_of(x));
        }
        // ---- off the chain ----
        facc += S1.y * (im_lane ? -pg : pg) * unext;              // dt_k Im(g conj(u_{k+1})): re lanes g_im u_re, im lanes -g_re u_im
        const float invp = SP0.y;
        const float yhp = PRV.y * invp, yhpp = PRV.yp * invp;
        const float uk = k > 0 ? rhp.x * yhp + sgn * rhp.y * yhpp : ps0;       // u_k = rho_{k-1} yhat_{k-1}  (psi_0 at k = 0)
        c3 = fmaf(SP0.w, PRV.h, -(yhp * (S1.x * SP0.z * invp)));               // the g-independent part of ybar_{k-1}
        const v4f nS0 = tab_row(km2, 0), nS1 = tab_row(km2, 1);
        CUR = row_at(k - 4);
        const float d = reduce_slices(dRe0, dIm0, dRe1, dIm1, clip1);
        accS += d * uk;
        g = yb + d;
        unext = uk;
        rh = rhp; rhp = nrh;
        S0 = SP0; S1 = SP1; SP0 = nS0; SP1 = nS1;
        p ^= 1;
    };
    for (int blk = (N - 1) / 4; blk >= 0; --blk) {
        const int kb = 4 * blk;
        if (kb + 3 < N) step(kb + 3, ring3, ring2);
        if (kb + 2 < N) step(kb + 2, ring2, ring1);
        if (kb + 1 < N) step(kb + 1, ring1, ring0);
        step(kb, ring0, ring3);
    }

    // ---- the pair's slab: f | psi0bar_re | psi0bar_im | A (the R / Q sections are written by k_grad_gemm) ----
    float* slab = P.slabs + (size_t)blockIdx.x * P.slab_floats;
    constexpr int DD = PD * PD;
    {
        float f = facc * wq, g0 = g * wq;
        f += wdpp<0x128>(f);                                      // + the other clip (lane ^ 8)
        g0 += wdpp<0x128>(g0);
        const float ft = f + partner16(f, im_lane);               // + the other component's share
        if (!clip1) {
            if (!im_lane) slab[4 * DD + row] = ft;
            slab[4 * DD + (im_lane ? 2 * PD : PD) + row] = P.gphi ? 0.f : g0;
        }
        if (P.gphi) P.gphi[((size_t)(2 * blockIdx.x + clip) * 2 + comp) * PD + row] = g * wq;    // every column's own cotangent
    }
    {
        float t = accS * wq;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
        float a = accA;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off, 64);
        if (lane == 0) redA[w] = -(a / (A * A)) - t / A;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) tot += redA[ww];
            slab[4 * DD + 3 * PD] = tot;
            slab[4 * DD + 3 * PD + 1] = 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ymx = fmaxf(ymx, __shfl_xor(ymx, off, 64));
        if (lane == 0) redA[w] = ymx;
        __syncthreads();
        if (threadIdx.x == 0) {
            float m = 0.f;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) m = fmaxf(m, redA[ww]);
            P.opmax[blockIdx.x] = m;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// gradient contraction: k_grad_gemm (cmps_grad_gemm.h) on this family's rows -- a stash vector is in lane order (position p = 64 (row
// / 16) + 8 q + (row % 8), q = (row half, component, clip)), and the reverse scan normalises with rsq_newton
// ------------------------------------------------------------------------------------------------------------------------
template <int PD>
struct WideRows {
    static __device__ __forceinline__ int y_off(int tid, int c) { return (((tid >> 4) << 5) | (tid & 15)) + 16 * c; }
    static __device__ __forceinline__ int yb_off(int tid, int c) { return y_off(tid, c); }
    static __device__ __forceinline__ float rsq(float m) { return rsq_newton(m); }
};

// ------------------------------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------------------------------
template <typename K>
static hipError_t wide_lds_attr(K kernel, size_t shm) {
    if (shm <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
}

template <int PD, bool LEGACY = false>
static hipError_t fwd_wide_t(const Dev& P, const float* audio, float* loss, bool save, bool hy_f16, bool chain_mfma, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    hipError_t e;
    if (save) {
        // the serial chain, then H y / e_k for all (clip, step) pairs as one GEMM launch, then the sequential loss sums
        const size_t shm = WideGeom<PD>::FWD_LDS_CHAIN, shm_hy = HyGeom<PD>::LDS;
        e = wide_lds_attr(k_fwd_wide<PD, true, LEGACY>, shm);
        if (e == hipSuccess) e = hy_f16 ? wide_lds_attr(k_hy_wide<PD, true>, shm_hy) : wide_lds_attr(k_hy_wide<PD, false>, shm_hy);
        if (e != hipSuccess) return e;
        if (chain_mfma && !LEGACY) {
            KScope ks("k_fwd_chain16", s);
            if ((e = launch_fwd_chain16(P, audio, s)) != hipSuccess) return e;
        } else {
            KScope ks("k_fwd_wide", s);
            hipLaunchKernelGGL((k_fwd_wide<PD, true, LEGACY>), dim3(nb), dim3(4 * PD), shm, s, P, audio, loss);
        }
        {
            KScope ks(hy_f16 ? "k_hy_wide<f16x2>" : "k_hy_wide<3>", s);
            const dim3 grid(nb, (unsigned)((P.N + HCHUNK - 1) / HCHUNK));
            if (hy_f16) hipLaunchKernelGGL((k_hy_wide<PD, true>), grid, dim3(2 * PD), shm_hy, s, P);
            else hipLaunchKernelGGL((k_hy_wide<PD, false>), grid, dim3(2 * PD), shm_hy, s, P);
        }
        { KScope ks("k_loss_wide", s); hipLaunchKernelGGL(k_loss_wide<LEGACY>, dim3((unsigned)P.B), dim3(64), 0, s, P, audio, loss); }
    } else {
        const size_t shm = WideGeom<PD>::FWD_LDS;
        e = wide_lds_attr(k_fwd_wide<PD, false, LEGACY>, shm);
        if (e != hipSuccess) return e;
        KScope ks("k_fwd_wide", s);
        hipLaunchKernelGGL((k_fwd_wide<PD, false, LEGACY>), dim3(nb), dim3(4 * PD), shm, s, P, audio, loss);
    }
    return hipGetLastError();
}

// the legacy AudioMPS arithmetic on the wide kernels (32 < D <= 128): the fp32 VALU chain kernels in their LEGACY mode + the same GEMMs
hipError_t launch_fwd_wide_legacy(const Dev& P, const float* audio, float* loss, bool save, bool hy_f16, hipStream_t s) {
    if (P.DP == 128) return fwd_wide_t<128, true>(P, audio, loss, save, hy_f16, false, s);
    if (P.DP == 96) return fwd_wide_t<96, true>(P, audio, loss, save, hy_f16, false, s);
    if (P.DP == 64) return fwd_wide_t<64, true>(P, audio, loss, save, hy_f16, false, s);
    return hipErrorInvalidValue;
}

hipError_t launch_bwd_wide_legacy(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    if (P.DP == 128) hipLaunchKernelGGL((k_bwd_wide<128, true>), dim3(nb), dim3(512), 0, s, P, audio);
    else if (P.DP == 96) hipLaunchKernelGGL((k_bwd_wide<96, true>), dim3(nb), dim3(384), 0, s, P, audio);
    else if (P.DP == 64) hipLaunchKernelGGL((k_bwd_wide<64, true>), dim3(nb), dim3(256), 0, s, P, audio);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_fwd_wide(const Dev& P, const float* audio, float* loss, bool save, bool hy_f16, bool chain_mfma, hipStream_t s) {
    if (P.DP == 128) return fwd_wide_t<128>(P, audio, loss, save, hy_f16, chain_mfma, s);
    if (P.DP == 96) return fwd_wide_t<96>(P, audio, loss, save, hy_f16, chain_mfma, s);
    if (P.DP == 64) return fwd_wide_t<64>(P, audio, loss, save, hy_f16, chain_mfma, s);
    return hipErrorInvalidValue;
}

hipError_t launch_sample_wide(const Dev& P, const float* noise, int n, int length, float* out, hipStream_t s) {
    const unsigned nb = (unsigned)((n + 1) / 2);
    if (P.DP == 128) hipLaunchKernelGGL(k_sample_wide<128>, dim3(nb), dim3(512), 0, s, P, noise, n, length, out);
    else if (P.DP == 96) hipLaunchKernelGGL(k_sample_wide<96>, dim3(nb), dim3(384), 0, s, P, noise, n, length, out);
    else if (P.DP == 64) hipLaunchKernelGGL(k_sample_wide<64>, dim3(nb), dim3(256), 0, s, P, noise, n, length, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_bwd_wide(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    if (P.DP == 128) hipLaunchKernelGGL(k_bwd_wide<128>, dim3(nb), dim3(512), 0, s, P, audio);
    else if (P.DP == 96) hipLaunchKernelGGL(k_bwd_wide<96>, dim3(nb), dim3(384), 0, s, P, audio);
    else if (P.DP == 64) hipLaunchKernelGGL(k_bwd_wide<64>, dim3(nb), dim3(256), 0, s, P, audio);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

template <int PD, int NPC, bool F16 = false, bool LEGACY = false>
static hipError_t grad_wide_t(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + 1) / 2);
    hipLaunchKernelGGL((k_grad_gemm<PD, NPC, WideRows<PD>, F16, LEGACY>), dim3(nb), dim3(2 * PD), 0, s, P, audio);   // static LDS: 2 x NPC x 160 PD + 4 KB
    return hipGetLastError();
}

// legacy mode: two fp16 pieces (CMPS_RANK1_F16X2 / DEFAULT) or three bf16 pieces (every other value)
hipError_t launch_grad_wide_legacy(const Dev& P, const float* audio, bool f16, hipStream_t s) {
    if (f16) {
        if (P.DP == 128) return grad_wide_t<128, 2, true, true>(P, audio, s);
        if (P.DP == 96) return grad_wide_t<96, 2, true, true>(P, audio, s);
        if (P.DP == 64) return grad_wide_t<64, 2, true, true>(P, audio, s);
    } else {
        if (P.DP == 128) return grad_wide_t<128, 3, false, true>(P, audio, s);
        if (P.DP == 96) return grad_wide_t<96, 3, false, true>(P, audio, s);
        if (P.DP == 64) return grad_wide_t<64, 3, false, true>(P, audio, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_grad_wide(const Dev& P, const float* audio, int pieces, hipStream_t s) {
    if (pieces == -2) {                                           // two fp16 pieces (CMPS_RANK1_F16X2)
        if (P.DP == 128) return grad_wide_t<128, 2, true>(P, audio, s);
        if (P.DP == 96) return grad_wide_t<96, 2, true>(P, audio, s);
        if (P.DP == 64) return grad_wide_t<64, 2, true>(P, audio, s);
    } else if (pieces == 3) {
        if (P.DP == 128) return grad_wide_t<128, 3>(P, audio, s);
        if (P.DP == 96) return grad_wide_t<96, 3>(P, audio, s);
        if (P.DP == 64) return grad_wide_t<64, 3>(P, audio, s);
    } else {
        if (P.DP == 128) return grad_wide_t<128, 2>(P, audio, s);
        if (P.DP == 96) return grad_wide_t<96, 2>(P, audio, s);
        if (P.DP == 64) return grad_wide_t<64, 2>(P, audio, s);
    }
    return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------------------------------
// RhoCMPS on the wide kernels (32 < D <= 128; the sections of RhoLayout::vrank > 0).  P carries the tables of cmps_set_params; every
// buffer of the virtual-clip run lives in the rho workspace.
// ------------------------------------------------------------------------------------------------------------------------
static Dev rho_virtual_dev(const Dev& P, const RhoDev& W) {
    Dev V = P;
    V.B = P.B * W.vrank;
    V.stash = reinterpret_cast<float2*>(W.vstash);
    V.hst = W.vstash;
    V.stash_layout = 4;
    V.scal = W.vscal;
    V.gops = W.vgops;
    V.opmax = W.vopmax;
    V.slabs = W.vslabs;
    V.sums = W.vsums;
    V.slab_floats = W.vslab_floats;
    V.phi0 = W.vphi;
    V.phi_rank = W.vrank;
    V.gphi = W.gphi;
    V.status = nullptr;
    return V;
}

template <int PD>
static hipError_t fwd_rho_wide_t(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool hy_f16, hipStream_t s) {
    const int npairs = W.vrank / 2, NC = (P.N + WCH - 1) / WCH;
    const size_t shm = rho_wide_lds(P.D, W.rank), shm_hy = HyGeom<PD>::LDS;
    hipError_t e = wide_lds_attr(k_fwd_wide_rho<PD>, shm);
    if (e == hipSuccess) e = hy_f16 ? wide_lds_attr(k_hy_wide<PD, true>, shm_hy) : wide_lds_attr(k_hy_wide<PD, false>, shm_hy);
    if (e != hipSuccess) return e;
    Dev V = rho_virtual_dev(P, W);
    {
        KScope ks("k_rho_expand_audio", s);
        hipLaunchKernelGGL(k_rho_pad_phi, dim3((unsigned)((W.vrank * PD + 255) / 256)), dim3(256), 0, s, W.phi0, W.vphi, W.rank, W.vrank, PD);
        hipLaunchKernelGGL(k_rho_expand_audio, dim3((unsigned)V.B), dim3(256), 0, s, audio, W.vaudio, P.T, W.vrank);
    }
    { KScope ks("k_fwd_wide_rho", s); hipLaunchKernelGGL(k_fwd_wide_rho<PD>, dim3((unsigned)P.B), dim3(4 * PD), shm, s, V, audio, W.rscal, npairs); }
    { KScope ks("k_rho_spread_n", s); hipLaunchKernelGGL(k_rho_spread_n, dim3((unsigned)V.B, (unsigned)NC), dim3(64), 0, s, W.rscal, W.vscal, NC, W.vrank); }
    {
        KScope ks(hy_f16 ? "k_hy_wide<f16x2>" : "k_hy_wide<3>", s);
        const dim3 grid((unsigned)(V.B / 2), (unsigned)((P.N + HCHUNK - 1) / HCHUNK));
        if (hy_f16) hipLaunchKernelGGL((k_hy_wide<PD, true>), grid, dim3(2 * PD), shm_hy, s, V);
        else hipLaunchKernelGGL((k_hy_wide<PD, false>), grid, dim3(2 * PD), shm_hy, s, V);
    }
    { KScope ks("k_rho_merge_e", s); hipLaunchKernelGGL(k_rho_merge_e, dim3((unsigned)P.B, (unsigned)NC), dim3(64), 0, s, W.rscal, W.vscal, NC, W.vrank); }
    {
        Dev R = P;
        R.scal = W.rscal;
        KScope ks("k_loss_wide", s);
        hipLaunchKernelGGL(k_loss_wide<false>, dim3((unsigned)P.B), dim3(64), 0, s, R, audio, loss);
    }
    return hipGetLastError();
}

hipError_t launch_fwd_rho_wide(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool hy_f16, hipStream_t s) {
    if (P.DP == 128) return fwd_rho_wide_t<128>(P, W, audio, loss, hy_f16, s);
    if (P.DP == 96) return fwd_rho_wide_t<96>(P, W, audio, loss, hy_f16, s);
    if (P.DP == 64) return fwd_rho_wide_t<64>(P, W, audio, loss, hy_f16, s);
    return hipErrorInvalidValue;
}

// reverse chain + gradient GEMM on the virtual clips, the common reduction, the closing terms (k_finalize), the columns' cotangents.
// pieces: -2 two fp16 pieces, 2 / 3 bf16 pieces (launch_grad_wide).  grad_out: the layout of cmps_rho_loss_bwd.
hipError_t launch_bwd_rho_wide(const Dev& P, const RhoDev& W, const float* loss, float* grad_out, int pieces, hipStream_t s) {
    Dev V = rho_virtual_dev(P, W);
    hipError_t e;
    { KScope ks("k_bwd_wide", s); e = launch_bwd_wide(V, W.vaudio, s); }
    if (e != hipSuccess) return e;
    { KScope ks("k_grad_gemm", s); e = launch_grad_wide(V, W.vaudio, pieces, s); }
    if (e != hipSuccess) return e;
    KScope ks("reduce + finalize", s);
    Dev Vp = V;
    Vp.B = V.B / 2;
    e = launch_reduce_only(Vp, s);
    if (e != hipSuccess) return e;
    Dev F = V;
    F.B = P.B;                                                    // the loss sum runs over the real clips
    F.abar_fix = 1;
    e = launch_finalize_only(F, loss, grad_out, s);
    if (e != hipSuccess) return e;
    const int D = P.D, n = W.rank * D;
    float* dphi = grad_out + (2 * D * D + 3 * D + 2);
    hipLaunchKernelGGL(k_rho_phi_reduce, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)W.gphi, P.B, W.vrank, W.rank, D, P.DP,
                       dphi, dphi + n);
    return hipGetLastError();
}

}  // namespace cmps
