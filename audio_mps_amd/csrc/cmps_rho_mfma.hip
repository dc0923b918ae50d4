// RhoCMPS forward scan on the matrix cores (D <= 32, rank <= 32): "per step a 64 x 64 x 2r real GEMM per clip".
//
// The density matrix is carried as its `rank` columns u_a (see cmps_rho.hip for the arithmetic and the reference lines,
// model.py:133-203).  All columns see the same step matrices, so a step is two real GEMMs over the ROW ARRAY
//     U [32 columns x 64 reals]   (row a = column u_a of rho, n = 2 i + {re, im}; rows >= rank are zero and stay zero):
//     Y  = U + U W_k ,   W_k = real 64 x 64 form of M_k = Q + s_k R         (model.py:172-187, the two products merged)
//     HY = Y W_H     ,   W_H = real form of R + R^dagger                     (model.py:189-196)
// each 2 tiles x 4 k-steps of v_mfma_f32_32x32x16_bf16 with both operands split EXACTLY into three bf16 pieces
// (8 + 8 + 8 significand bits) and six products per pair accumulated in fp32 (24 operand bits, fp32-faithful products): the
// routine of the pure-state forward's loss wave (cmps_wave2.hip) with rows = columns of rho instead of time steps.
// e = sum_a y_a . (H y_a) and n = tr rho' = sum_a |y_a|^2 are whole-array sums; the rotation u_a' = rho_k (.) y_a / sqrt(n)
// is lane-local in the C/D layout (a component's partner sits in the neighbouring lane: one DPP move).
// One wavefront per clip; the row arrays live in LDS between steps (padded rows: conflict-free 32-row reads).
// Stash (layout 2): per (step, column a < rank) one 512-B row of 64 pairs (y_a[n], (H y_a)[n]).
#include "cmps_wave_util.h"

namespace cmps {

namespace {

constexpr int RRLD = 68;      // floats per row of the LDS row arrays: 64 + 4 of padding

// exact three-way bf16 split of two floats (even element in the low half of every packed word)
__device__ __forceinline__ void split3(float fe, float fo, unsigned& H, unsigned& M, unsigned& L) {
    const unsigned xe = __float_as_uint(fe), xo = __float_as_uint(fo);
    H = __builtin_amdgcn_perm(xo, xe, 0x07060302u);
    const float re = fe - __uint_as_float(xe & 0xFFFF0000u), ro = fo - __uint_as_float(xo & 0xFFFF0000u);
    const unsigned me = __float_as_uint(re), mo = __float_as_uint(ro);
    M = __builtin_amdgcn_perm(mo, me, 0x07060302u);
    const float le = re - __uint_as_float(me & 0xFFFF0000u), lo = ro - __uint_as_float(mo & 0xFFFF0000u);
    L = __builtin_amdgcn_perm(__float_as_uint(lo), __float_as_uint(le), 0x07060302u);   // <= 8 bits left: exact
}
__device__ __forceinline__ bf8 frag4(const unsigned (&f)[4]) { return __builtin_bit_cast(bf8, v4u{f[0], f[1], f[2], f[3]}); }

// acc += A B over the six piece pairs (hi,hi) (hi,mid) (mid,hi) (hi,lo) (lo,hi) (mid,mid)
__device__ __forceinline__ void mfma6(v16f& acc, const unsigned (&AH)[4], const unsigned (&AM)[4], const unsigned (&AL)[4],
                                      const unsigned (&BH)[4], const unsigned (&BM)[4], const unsigned (&BL)[4]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(AH), frag4(BH), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(AH), frag4(BM), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(AM), frag4(BH), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(AH), frag4(BL), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(AL), frag4(BH), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag4(AM), frag4(BM), acc, 0, 0, 0);
}

// ---- round 4: two fp16 pieces per value (round to nearest: hi = f16(x), lo = f16(x - hi), 11 + 1 + 11 + 1 bits) and three products
// on v_mfma_f32_32x32x16_f16 instead of three bf16 pieces and six: the accuracy class of the split above at half the MFMAs and
// 6 instead of 11 VALU per pair of values (cmps_grad_gemm.h).  fp16 has 5 exponent bits: the caller scales every
// operand class by a power of two from a guaranteed bound and unscales the accumulators.
typedef _Float16 hf8r __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(float fe, float fo, unsigned& H, unsigned& L) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 hh = h2{(_Float16)fe, (_Float16)fo};
    H = __builtin_bit_cast(unsigned, hh);
    const float re = fe - (float)hh.x, ro = fo - (float)hh.y;
    L = __builtin_bit_cast(unsigned, h2{(_Float16)re, (_Float16)ro});
}
__device__ __forceinline__ hf8r frag4h(const unsigned (&f)[4]) { return __builtin_bit_cast(hf8r, v4u{f[0], f[1], f[2], f[3]}); }
// acc += A B over the piece pairs (hi,hi) (hi,lo) (lo,hi)
__device__ __forceinline__ void mfma3(v16f& acc, const unsigned (&AH)[4], const unsigned (&AL)[4], const unsigned (&BH)[4], const unsigned (&BL)[4]) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag4h(AH), frag4h(BH), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag4h(AH), frag4h(BL), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag4h(AL), frag4h(BH), acc, 0, 0, 0);
}
__device__ __forceinline__ void pieces8h(const float (&x)[16], int s2, float sc, unsigned (&H)[4], unsigned (&L)[4]) {
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) split2h(x[8 * s2 + 2 * e2] * sc, x[8 * s2 + 2 * e2 + 1] * sc, H[e2], L[e2]);
}
// the largest power of two S with bound S < 2^target (exponent clamped: S and 1 / S stay normal); 1 / S for such an S
__device__ __forceinline__ float pow2_below(float bound, int target) {
    int se = target - ((int)((__float_as_uint(bound) >> 23) & 0xFFu) - 126);
    se = se > 60 ? 60 : se < -60 ? -60 : se;
    return __uint_as_float((unsigned)(127 + se) << 23);
}
__device__ __forceinline__ float pow2_recip(float s) { return __uint_as_float(0x7F000000u - __float_as_uint(s)); }
__device__ __forceinline__ float max64(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
    return x;
}

// W[m][n] of a complex 32 x 32 matrix M acting on (re, im)-interleaved vectors: (M y)[n = 2 i + c] = sum_m y[m = 2 j + c'] W[m][n],
// W = Mr_ij for c' = c, -Mi_ij for (c', c) = (1, 0), +Mi_ij for (0, 1)
__device__ __forceinline__ float wform(float2 mij, int cp, int cc) { return cp == cc ? mij.x : (cc ? mij.y : -mij.y); }

// pieces of eight registers x[8 s2 .. 8 s2 + 7] of a C/D tile as the K-fragment of k-step s2
__device__ __forceinline__ void pieces8(const float (&x)[16], int s2, unsigned (&H)[4], unsigned (&M)[4], unsigned (&L)[4]) {
#pragma unroll
    for (int e2 = 0; e2 < 4; ++e2) split3(x[8 * s2 + 2 * e2], x[8 * s2 + 2 * e2 + 1], H[e2], M[e2], L[e2]);
}
__device__ __forceinline__ float dpp_nb(float x) {      // the value of lane n ^ 1
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0xB1, 0xf, 0xf, true));
}

}  // namespace

// GRAD1: also accumulate the forward's part of Rbar, sum_k 2 ebar_k Y^T Y (RhoDev::p1), which k_bwd_rho_mfma adds to its own sums; the
// virtual-clip reverse sweep (round 5, k_bwd_wave per column: cmps_rho_wave.hip) forms all three rank-1 sums itself and runs without it
template <bool SAVE, bool F16, bool GRAD1 = true>
__global__ __launch_bounds__(64 * WAVES, 1) void k_fwd_rho_mfma(Dev P, RhoDev W, const float* __restrict__ audio,
                                                                float* __restrict__ loss_out) {
    __shared__ __attribute__((aligned(16))) float Urow[WAVES][32 * RRLD];
    __shared__ __attribute__((aligned(16))) float Yrow[WAVES][32 * RRLD];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH, r = W.rank;
    const int col = lane & 31, hk = lane >> 5;
    float* U = &Urow[w][0];
    float* Y = &Yrow[w][0];
    // ---- constant operands: W forms of Q and R (fp32, merged per step) and the bf16 pieces of the W form of R + R^dagger;
    // lane (col, hk) holds W[16 ks + 8 hk + e][32 t + col], e = 0..7, for tile t and k-step ks
    float WQ[2][4][8], WR[2][4][8];
    float HWraw[F16 ? 2 : 1][F16 ? 4 : 1][F16 ? 8 : 1];            // (F16: H's real form until its scale is known)
    unsigned HH[2][4][4], HM[F16 ? 1 : 2][4][4], HL[2][4][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * t + col, ii = n >> 1, cc = n & 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float wh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = 16 * ks + 8 * hk + e, jj = m >> 1, cp = m & 1;
                const float2 rij = P.R[ii * DPW + jj], rji = P.RT[ii * DPW + jj], qij = P.Q[ii * DPW + jj];
                WQ[t][ks][e] = wform(qij, cp, cc);
                WR[t][ks][e] = wform(rij, cp, cc);
                wh[e] = wform(make_float2(rij.x + rji.x, rij.y - rji.y), cp, cc);      // (R + R^dagger)[ii][jj]
            }
            if constexpr (!F16) {
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) split3(wh[2 * e2], wh[2 * e2 + 1], HH[t][ks][e2], HM[t][ks][e2], HL[t][ks][e2]);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) HWraw[t][ks][e] = wh[e];
            }
        }
    }
    // F16: power-of-two scales from guaranteed bounds.  U: the columns of rho, tr rho = 1 => |entries| <= 1 (2^13).  W_k = W_Q + s_k W_R:
    // max |W_Q| + max |s| max |W_R| over the clip (2^15; the scale is folded into W_Q, W_R once).  Y = U + U W_k: rows of norm
    // <= 1 + |W_Q|_F + max |s| |W_R|_F (2^13).  H = R + R^dagger: its largest entry (2^15).
    float sU = 1.f, sW = 1.f, sY = 1.f, sH = 1.f, iUW = 1.f, iYH = 1.f, iYY = 1.f;
    if constexpr (F16) {
        float mq = 0.f, mr = 0.f, mh = 0.f, fq = 0.f, fr = 0.f, ms = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    mq = fmaxf(mq, fabsf(WQ[t][ks][e])); mr = fmaxf(mr, fabsf(WR[t][ks][e])); mh = fmaxf(mh, fabsf(HWraw[t][ks][e]));
                    fq = fmaf(WQ[t][ks][e], WQ[t][ks][e], fq); fr = fmaf(WR[t][ks][e], WR[t][ks][e], fr);
                }
        const float* xr = audio + (size_t)b * T;
        for (int idx = lane; idx < N; idx += 64) ms = fmaxf(ms, fabsf((xr[idx + 1] - xr[idx]) / dev_A(P)));
        mq = max64(mq); mr = max64(mr); mh = max64(mh); ms = 1.001f * max64(ms);
        const float nq = 1.001f * sqrtf(sum64(fq)), nr = 1.001f * sqrtf(sum64(fr));
        sU = 8192.f;
        sW = pow2_below(mq + ms * mr, 15);
        sY = pow2_below(1.01f * (1.0f + nq + ms * nr), 13);
        sH = pow2_below(mh, 15);
        iUW = pow2_recip(sU) * pow2_recip(sW); iYH = pow2_recip(sY) * pow2_recip(sH); iYY = pow2_recip(sY) * pow2_recip(sY);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { WQ[t][ks][e] *= sW; WR[t][ks][e] *= sW; }
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) split2h(HWraw[t][ks][2 * e2] * sH, HWraw[t][ks][2 * e2 + 1] * sH, HH[t][ks][e2], HL[t][ks][e2]);
            }
    }
    // ---- initial columns: rows a < rank of U = phi_a (model.py:127-136), the rest zero
    for (int a = 0; a < 32; ++a) {
        float v = 0.f;
        if (a < r) {
            const float2 p = W.phi0[a * DPW + (lane >> 1)];
            v = (lane & 1) ? p.y : p.x;
        }
        U[a * RRLD + lane] = v;
    }
    const float* xrow = audio + (size_t)b * T;
    float2* st = SAVE ? reinterpret_cast<float2*>(W.stash) + (size_t)b * N * r * 64 : nullptr;
    float* sc = SAVE ? W.scal + (size_t)b * NC * 128 : nullptr;
    const float A = dev_A(P);
    const float sgn = (col & 1) ? 1.f : -1.f;                  // Im lanes add rho_y * partner, Re lanes subtract it
    // the part of the gradient that needs no cotangent: P1 = sum_k 2 ebar_k Y^T Y (real form of sum_k 2 ebar_k sum_a y_a y_a^dagger),
    // a GEMM over the rows whose operands are the C/D tiles themselves (see k_bwd_rho_mfma)
    v16f P1[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int z = 0; z < 2; ++z) P1[x][z] = v16f{};
    float loss = 0.f;
    for (int c = 0; c < NC; ++c) {
        const int kbeg = c * CH;
        const int cnt = (N - kbeg) < CH ? (N - kbeg) : CH;
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f, x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        const float incv = x1 - x0;                              // model.py:138
        const float sv = incv / A;                               // :175
        float nvec = 1.f, evec = 0.f;
        for (int kk = 0; kk < cnt; ++kk) {
            const int k = kbeg + kk;
            const float s = rdlane(sv, kk);
            // the rotation of this step, for this lane's component of either tile (L2-resident table)
            const float2 rh0 = P.rho[(size_t)k * DPW + (col >> 1)], rh1 = P.rho[(size_t)k * DPW + 16 + (col >> 1)];
            // ---- U W_k ----
            v16f a0 = {}, a1 = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4 f0 = *reinterpret_cast<const float4*>(U + col * RRLD + 16 * ks + 8 * hk);
                const float4 f1 = *reinterpret_cast<const float4*>(U + col * RRLD + 16 * ks + 8 * hk + 4);
                unsigned AH[4], AM[4], AL[4];
                if constexpr (F16) {
                    split2h(f0.x * sU, f0.y * sU, AH[0], AL[0]);
                    split2h(f0.z * sU, f0.w * sU, AH[1], AL[1]);
                    split2h(f1.x * sU, f1.y * sU, AH[2], AL[2]);
                    split2h(f1.z * sU, f1.w * sU, AH[3], AL[3]);
                } else {
                    split3(f0.x, f0.y, AH[0], AM[0], AL[0]);
                    split3(f0.z, f0.w, AH[1], AM[1], AL[1]);
                    split3(f1.x, f1.y, AH[2], AM[2], AL[2]);
                    split3(f1.z, f1.w, AH[3], AM[3], AL[3]);
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    unsigned BH[4], BM[4], BL[4];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const float we = fmaf(s, WR[t][ks][2 * e2], WQ[t][ks][2 * e2]), wo = fmaf(s, WR[t][ks][2 * e2 + 1], WQ[t][ks][2 * e2 + 1]);
                        if constexpr (F16) split2h(we, wo, BH[e2], BL[e2]);       // (W_Q, W_R carry the scale sW)
                        else split3(we, wo, BH[e2], BM[e2], BL[e2]);
                    }
                    if constexpr (F16) mfma3(t ? a1 : a0, AH, AL, BH, BL);
                    else mfma6(t ? a1 : a0, AH, AM, AL, BH, BM, BL);
                }
            }
            // ---- Y = U + U W_k in the C/D layout: column n = 32 t + col, rows (r & 3) + 8 (r >> 2) + 4 hk ----
            float y0[16], y1[16];
            float accn = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hk;
                y0[q] = F16 ? fmaf(a0[q], iUW, U[row * RRLD + col]) : U[row * RRLD + col] + a0[q];
                y1[q] = F16 ? fmaf(a1[q], iUW, U[row * RRLD + 32 + col]) : U[row * RRLD + 32 + col] + a1[q];
                accn = fmaf(y0[q], y0[q], accn);
                accn = fmaf(y1[q], y1[q], accn);
                Y[row * RRLD + col] = y0[q];
                Y[row * RRLD + 32 + col] = y1[q];
            }
            // ---- H Y ----
            v16f h0 = {}, h1 = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4 f0 = *reinterpret_cast<const float4*>(Y + col * RRLD + 16 * ks + 8 * hk);
                const float4 f1 = *reinterpret_cast<const float4*>(Y + col * RRLD + 16 * ks + 8 * hk + 4);
                unsigned AH[4], AM[4], AL[4];
                if constexpr (F16) {
                    split2h(f0.x * sY, f0.y * sY, AH[0], AL[0]);
                    split2h(f0.z * sY, f0.w * sY, AH[1], AL[1]);
                    split2h(f1.x * sY, f1.y * sY, AH[2], AL[2]);
                    split2h(f1.z * sY, f1.w * sY, AH[3], AL[3]);
                    mfma3(h0, AH, AL, HH[0][ks], HL[0][ks]);
                    mfma3(h1, AH, AL, HH[1][ks], HL[1][ks]);
                } else {
                    split3(f0.x, f0.y, AH[0], AM[0], AL[0]);
                    split3(f0.z, f0.w, AH[1], AM[1], AL[1]);
                    split3(f1.x, f1.y, AH[2], AM[2], AL[2]);
                    split3(f1.z, f1.w, AH[3], AM[3], AL[3]);
                    mfma6(h0, AH, AM, AL, HH[0][ks], HM[0][ks], HL[0][ks]);
                    mfma6(h1, AH, AM, AL, HH[1][ks], HM[1][ks], HL[1][ks]);
                }
            }
            if constexpr (F16 && SAVE) {                         // the stash keeps H y itself
#pragma unroll
                for (int q = 0; q < 16; ++q) { h0[q] *= iYH; h1[q] *= iYH; }
            }
            float acce = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acce = fmaf(y0[q], h0[q], acce);
                acce = fmaf(y1[q], h1[q], acce);
            }
            const float n = sum64(accn);                         // tr rho', :200
            const float e = (F16 && !SAVE) ? sum64(acce) * iYH : sum64(acce);   // Re tr(x rho'), :195-196
            nvec = lane == kk ? n : nvec;
            evec = lane == kk ? e : evec;
            if (SAVE) {
                // register q holds row s0 = (q & 3) + 8 (q >> 2) in lanes 0-31 and row s0 + 4 in lanes 32-63, for the two tiles:
                // one v_permlane32_swap between the tiles puts a whole 512-B row into each store
                float2* sbase = st + (size_t)k * r * 64 + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int s0 = (q & 3) + 8 * (q >> 2);
                    const auto ys = __builtin_amdgcn_permlane32_swap(__float_as_uint(y0[q]), __float_as_uint(y1[q]), false, false);
                    const auto hs = __builtin_amdgcn_permlane32_swap(__float_as_uint(h0[q]), __float_as_uint(h1[q]), false, false);
                    if (s0 < r) sbase[(size_t)s0 * 64] = make_float2(__uint_as_float(ys[0]), __uint_as_float(hs[0]));
                    if (s0 + 4 < r) sbase[(size_t)(s0 + 4) * 64] = make_float2(__uint_as_float(ys[1]), __uint_as_float(hs[1]));
                }
            }
            if (SAVE && GRAD1) {
                // 2 ebar_k with the reverse scan's own operations (cmps_rho_wave.hip: zbv, tev)
                const float inck = rdlane(incv, kk);
                const float zb = -1.0f / (1.0f + (e * inck) / A);
                const float te = 2.0f * (zb * inck / A);
                if constexpr (F16) {
                    // P1 += te (Y^T Y): the product of the step alone into fresh tiles (both operands are pieces of Y: split once), then
                    // one scaled add per accumulator register -- te never meets an fp16 piece
                    unsigned YH[2][2][4], YL[2][2][4];             // [tile][k-step s2]
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) { pieces8h(y0, s2, sY, YH[0][s2], YL[0][s2]); pieces8h(y1, s2, sY, YH[1][s2], YL[1][s2]); }
                    const float tes = te * iYY;
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                        for (int tb = 0; tb < 2; ++tb) {
                            v16f tmp = {};
                            mfma3(tmp, YH[ta][0], YL[ta][0], YH[tb][0], YL[tb][0]);
                            mfma3(tmp, YH[ta][1], YL[ta][1], YH[tb][1], YL[tb][1]);
#pragma unroll
                            for (int q = 0; q < 16; ++q) P1[ta][tb][q] = fmaf(tes, tmp[q], P1[ta][tb][q]);
                        }
                } else {
                float ty0[16], ty1[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) { ty0[q] = te * y0[q]; ty1[q] = te * y1[q]; }
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    unsigned TH[2][4], TM[2][4], TL[2][4];
                    pieces8(ty0, s2, TH[0], TM[0], TL[0]);
                    pieces8(ty1, s2, TH[1], TM[1], TL[1]);
#pragma unroll
                    for (int tb = 0; tb < 2; ++tb) {
                        unsigned VH[4], VM[4], VL[4];
                        pieces8(tb ? y1 : y0, s2, VH, VM, VL);
#pragma unroll
                        for (int ta = 0; ta < 2; ++ta) mfma6(P1[ta][tb], TH[ta], TM[ta], TL[ta], VH, VM, VL);
                    }
                }
                }
            }
            // ---- u_a' = rho_k (.) y_a / sqrt(n): the partner component sits in the neighbouring lane ----
            const float sc1 = sqrtf(1.0f / fmaxf(n, 1e-12f));    // :201 (columns scale with the square root)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hk;
                const float p0 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y0[q]), 0xB1, 0xf, 0xf, true));
                const float p1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(y1[q]), 0xB1, 0xf, 0xf, true));
                U[row * RRLD + col] = sc1 * fmaf(sgn * rh0.y, p0, rh0.x * y0[q]);
                U[row * RRLD + 32 + col] = sc1 * fmaf(sgn * rh1.y, p1, rh1.x * y1[q]);
            }
        }
        const float z = (evec * incv) / A;                       // :166
        const float lv = -logf(1.0f + z);
        for (int j = 0; j < cnt; ++j) loss += rdlane(lv, j);     // :155, sequential in time
        if (SAVE) {
            sc[(size_t)c * 128 + lane] = nvec;
            sc[(size_t)c * 128 + 64 + lane] = evec;
        }
    }
    if (lane == 0) loss_out[b] = loss;
    if (SAVE && GRAD1) {
        float* p1 = W.p1 + (size_t)b * 4096 + lane;
#pragma unroll
        for (int ta = 0; ta < 2; ++ta)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int q = 0; q < 16; ++q) p1[((ta * 2 + tb) * 16 + q) * 64] = P1[ta][tb][q];
    }
}

// ------------------------------------------------------------------------------------------------
// Reverse scan in the same row-array form (see k_bwd_rho_wave for the column-by-column statement of the adjoint).
// Arrays are [32 rows a][64 reals n]; between steps they live in registers in the C/D layout of the 32x32 MFMA (column n on
// the lane, 16 rows per lane and tile), where complex arithmetic needs only the neighbouring lane (n ^ 1, one DPP move):
//   yhat = y / sqrt(n_k);  u_{k+1} = rho_k yhat;  fbar += dt_k Im(g conj(u_{k+1}));  yhb = conj(rho_k) g;
//   dot = sum yhat . yhb;  ybar = (yhb - yhat dot) / sqrt(n_k) + 2 ebar_k (H y);   g <- ybar + ybar W'_k,  W'_k = form of Q + s_k R^dagger
// (one GEMM, operands split into bf16 x 3 as in the forward).  The gradient sums are GEMMs that contract over the ROWS a:
//   P2 += Ybar^T U_k,   P3 += (s_k Ybar)^T U_k,   P1 += (2 ebar_k Y)^T Y          (real 64 x 64 forms)
// and a C/D-layout tile IS the operand of such a product, for both factors, with no lane movement (registers 8 s .. 8 s + 7
// are the elements of k-step s in a permuted but COMMON row order); Qbar, Rbar are read off the real forms at the end.
// sum_k x_k Re(u^dagger R^dagger ybar) for dA comes from the merged product as in the pure-state kernels (Dev::abar_fix).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES, 1) void k_bwd_rho_mfma(Dev P, RhoDev W, const float* __restrict__ audio) {
    __shared__ __attribute__((aligned(16))) float Brow[WAVES][32 * RRLD];     // ybar rows: A operand of the W' GEMM
    __shared__ __attribute__((aligned(16))) float WQs[64 * 64], WDs[64 * 64]; // W forms of Q and R^dagger: [(t*4+ks)*8+e][lane]
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int col = lane & 31, hk = lane >> 5;
    // every wave fills its share of the constant operands (before any wave may leave)
    for (int v = w; v < 64; v += WAVES) {
        const int t = v >> 5, ks = (v >> 3) & 3, e = v & 7;
        const int n = 32 * t + col, ii = n >> 1, cc = n & 1;
        const int m = 16 * ks + 8 * hk + e, jj = m >> 1, cp = m & 1;
        const float2 qij = P.Q[ii * DPW + jj], rji = P.RT[ii * DPW + jj];           // R[jj][ii]
        WQs[v * 64 + lane] = wform(qij, cp, cc);
        WDs[v * 64 + lane] = wform(make_float2(rji.x, -rji.y), cp, cc);             // (R^dagger)[ii][jj]
    }
    __syncthreads();
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH, r = W.rank;
    float* Bw = &Brow[w][0];
    const float* xrow = audio + (size_t)b * T;
    const float2* st = reinterpret_cast<const float2*>(W.stash) + (size_t)b * N * r * 64;
    const float* sc = W.scal + (size_t)b * NC * 128;
    const float A = dev_A(P);
    const float sgn = (col & 1) ? 1.f : -1.f;        // rotation by rho: own*rho_x + sgn*partner*rho_y (conj: -sgn)
    // rows of this lane: a(q) = (q & 3) + 8 (q >> 2) + 4 hk
    // SEL 0: y_a(k) (.x of the stash pairs), SEL 1: (H y)_a(k); separate loads so that neither is held longer than it is used
    auto load_rows = [&](int k, int sel, float (&v0)[16], float (&v1)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int a = (q & 3) + 8 * (q >> 2) + 4 * hk;
            float t0 = 0.f, t1 = 0.f;
            if (a < r) {
                const float* row = reinterpret_cast<const float*>(st + ((size_t)k * r + a) * 64) + sel;
                t0 = row[2 * col];
                t1 = row[2 * (32 + col)];
            }
            v0[q] = t0; v1[q] = t1;
        }
    };
    v16f P2[2][2], P3[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int z = 0; z < 2; ++z) { P2[x][z] = v16f{}; P3[x][z] = v16f{}; }
    float g0[16], g1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { g0[q] = 0.f; g1[q] = 0.f; }
    float facc0 = 0.f, facc1 = 0.f, accA = 0.f, accS = 0.f;
    float y0[16], y1[16];                            // y(k) on entry to step k; y(k-1) from the middle of the step on
    load_rows(N - 1, 0, y0, y1);
    for (int c = NC - 1; c >= 0; --c) {
        const int kbeg = c * CH;
        const int cnt = (N - kbeg) < CH ? (N - kbeg) : CH;
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f, x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        const float incv = x1 - x0;
        const float nv = idx < N ? sc[(size_t)c * 128 + lane] : 1.f;
        const float ev = idx < N ? sc[(size_t)c * 128 + 64 + lane] : 0.f;
        const float invv = sqrtf(1.0f / fmaxf(nv, 1e-12f));
        const float okv = nv > 1e-12f ? 1.f : 0.f;
        const float exv = ev * incv;
        const float zv = exv / A;
        const float zbv = -1.0f / (1.0f + zv);
        const float tev = 2.0f * (zbv * incv / A);
        const float sv = incv / A;
        const float dtv = idx < N ? P.dtk[idx] : 0.f;
        if (idx < N) accA += zbv * (-exv / (A * A));
        const float nbelow = kbeg > 0 ? sc[(size_t)(c - 1) * 128 + 63] : 1.f;          // tr rho' of the step below the chunk
        const float invbelow = sqrtf(1.0f / fmaxf(nbelow, 1e-12f));
        for (int kk = cnt - 1; kk >= 0; --kk) {
            const int k = kbeg + kk;
            const float s = rdlane(sv, kk), inv = rdlane(invv, kk), ok = rdlane(okv, kk), te = rdlane(tev, kk);
            const float dtk = rdlane(dtv, kk);
            const float invp = kk > 0 ? rdlane(invv, kk - 1) : invbelow;
            const float2 rh0 = P.rho[(size_t)k * DPW + (col >> 1)], rh1 = P.rho[(size_t)k * DPW + 16 + (col >> 1)];
            const int kp = k > 0 ? k - 1 : 0;
            const float2 rp0 = P.rho[(size_t)kp * DPW + (col >> 1)], rp1 = P.rho[(size_t)kp * DPW + 16 + (col >> 1)];
            float hy0[16], hy1[16];
            load_rows(k, 1, hy0, hy1);                                    // lands during the first loop of phase A
            // ---- phase A: ybar ----
            float yb0[16], yb1[16];
            float accd = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float yh0 = y0[q] * inv, yh1 = y1[q] * inv;
                const float gp0 = dpp_nb(g0[q]), gp1 = dpp_nb(g1[q]);
                const float un0 = fmaf(sgn * rh0.y, dpp_nb(yh0), rh0.x * yh0);           // u_a(k+1) = rho_k yhat
                const float un1 = fmaf(sgn * rh1.y, dpp_nb(yh1), rh1.x * yh1);
                facc0 = fmaf(-sgn * dtk * gp0, un0, facc0);                              // Im(g conj(u)): +gi ur on Re lanes, -gr ui on Im lanes
                facc1 = fmaf(-sgn * dtk * gp1, un1, facc1);
                const float t0 = fmaf(-sgn * rh0.y, gp0, rh0.x * g0[q]);                 // conj(rho_k) g
                const float t1 = fmaf(-sgn * rh1.y, gp1, rh1.x * g1[q]);
                accd = fmaf(yh0, t0, accd);
                accd = fmaf(yh1, t1, accd);
                yb0[q] = t0;
                yb1[q] = t1;
            }
            const float dot = ok * sum64(accd);
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hk;
                yb0[q] = fmaf(te, hy0[q], (yb0[q] - (y0[q] * inv) * dot) * inv);
                yb1[q] = fmaf(te, hy1[q], (yb1[q] - (y1[q] * inv) * dot) * inv);
                Bw[row * RRLD + col] = yb0[q];
                Bw[row * RRLD + 32 + col] = yb1[q];
            }
            if (k > 0) load_rows(k - 1, 0, y0, y1);                       // in flight during the GEMM
            // ---- ybar W'_k ----
            v16f m0 = {}, m1 = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4 f0 = *reinterpret_cast<const float4*>(Bw + col * RRLD + 16 * ks + 8 * hk);
                const float4 f1 = *reinterpret_cast<const float4*>(Bw + col * RRLD + 16 * ks + 8 * hk + 4);
                unsigned AH[4], AM[4], AL[4];
                split3(f0.x, f0.y, AH[0], AM[0], AL[0]);
                split3(f0.z, f0.w, AH[1], AM[1], AL[1]);
                split3(f1.x, f1.y, AH[2], AM[2], AL[2]);
                split3(f1.z, f1.w, AH[3], AM[3], AL[3]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    unsigned BH[4], BM[4], BL[4];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2) {
                        const int v = (t * 4 + ks) * 8 + 2 * e2;
                        split3(fmaf(s, WDs[v * 64 + lane], WQs[v * 64 + lane]),
                               fmaf(s, WDs[(v + 1) * 64 + lane], WQs[(v + 1) * 64 + lane]), BH[e2], BM[e2], BL[e2]);
                    }
                    mfma6(t ? m1 : m0, AH, AM, AL, BH, BM, BL);
                }
            }
            // ---- u_a(k) = rho_{k-1} y_a(k-1) / sqrt(tr)  (the initial column at k = 0); g <- ybar + M' ybar ----
            float u0[16], u1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (k > 0) {
                    const float yp0 = y0[q] * invp, yp1 = y1[q] * invp;
                    u0[q] = fmaf(sgn * rp0.y, dpp_nb(yp0), rp0.x * yp0);
                    u1[q] = fmaf(sgn * rp1.y, dpp_nb(yp1), rp1.x * yp1);
                } else {
                    const int a = (q & 3) + 8 * (q >> 2) + 4 * hk;
                    float v0 = 0.f, v1 = 0.f;
                    if (a < r) {
                        const float2 p0 = W.phi0[a * DPW + (col >> 1)], p1 = W.phi0[a * DPW + 16 + (col >> 1)];
                        v0 = (col & 1) ? p0.y : p0.x;
                        v1 = (col & 1) ? p1.y : p1.x;
                    }
                    u0[q] = v0;
                    u1[q] = v1;
                }
                accS = fmaf(m0[q], u0[q], accS);
                accS = fmaf(m1[q], u1[q], accS);
                g0[q] = yb0[q] + m0[q];
                g1[q] = yb1[q] + m1[q];
            }
            // ---- gradient GEMMs over the rows a (ybar is read back from its LDS rows: it is not kept in registers) ----
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                unsigned YH[2][4], YM[2][4], YL[2][4], SH[2][4], SM[2][4], SL[2][4];
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    const int qa = 8 * s2 + 2 * e2, qb = qa + 1;
                    const int ra = (qa & 3) + 8 * (qa >> 2) + 4 * hk, rb = (qb & 3) + 8 * (qb >> 2) + 4 * hk;
                    const float a0 = Bw[ra * RRLD + col], b0 = Bw[rb * RRLD + col];
                    const float a1 = Bw[ra * RRLD + 32 + col], b1 = Bw[rb * RRLD + 32 + col];
                    split3(a0, b0, YH[0][e2], YM[0][e2], YL[0][e2]);
                    split3(a1, b1, YH[1][e2], YM[1][e2], YL[1][e2]);
                    split3(s * a0, s * b0, SH[0][e2], SM[0][e2], SL[0][e2]);
                    split3(s * a1, s * b1, SH[1][e2], SM[1][e2], SL[1][e2]);
                }
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) {
                    unsigned UH[4], UM[4], UL[4];
                    pieces8(tb ? u1 : u0, s2, UH, UM, UL);
#pragma unroll
                    for (int ta = 0; ta < 2; ++ta) {
                        mfma6(P2[ta][tb], YH[ta], YM[ta], YL[ta], UH, UM, UL);
                        mfma6(P3[ta][tb], SH[ta], SM[ta], SL[ta], UH, UM, UL);
                    }
                }
            }
        }
    }
    // ---- per-clip slab (layout of k_bwd_rho: the pure-state slab followed by the column cotangents) ----
    float* slab = W.slabs + (size_t)b * W.slab_floats;
    constexpr int DD = DPW * DPW;
    for (int idx = lane; idx < (int)W.slab_floats; idx += 64) slab[idx] = 0.f;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // complex 32 x 32 matrices from the real forms: tile (ta, tb) of P holds P[n = 32 ta + row(q)][n' = 32 tb + col];
    //   Z_ij = sum z_i conj(w_j) = (P[2i][2j] + P[2i+1][2j+1]) + i (P[2i+1][2j] - P[2i][2j+1]): rows 2i, 2i+1 are registers q, q+1,
    //   columns 2j, 2j+1 are neighbouring lanes; the even lanes write
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const int n = 32 * ta + (q & 3) + 8 * (q >> 2) + 4 * hk;     // even
                const int i = n >> 1, j = (32 * tb + col) >> 1;
                const float* p1 = W.p1 + (size_t)b * 4096 + ((ta * 2 + tb) * 16 + q) * 64 + lane;      // the forward's part
                const float r_e = p1[0] + P3[ta][tb][q], r_o = p1[64] + P3[ta][tb][q + 1];
                const float q_e = P2[ta][tb][q], q_o = P2[ta][tb][q + 1];
                const float rpe = dpp_nb(r_e), rpo = dpp_nb(r_o), qpe = dpp_nb(q_e), qpo = dpp_nb(q_o);
                if ((col & 1) == 0) {
                    slab[i * DPW + j] = r_e + rpo;
                    slab[DD + i * DPW + j] = r_o - rpe;
                    slab[2 * DD + i * DPW + j] = q_e + qpo;
                    slab[3 * DD + i * DPW + j] = q_o - qpe;
                }
            }
    // fbar_i: sum over the rows of this lane (both halves) and over the lane pair of component i
    {
        const float f0 = swapadd(facc0, facc0), f1 = swapadd(facc1, facc1);           // + the other half's rows
        const float t0 = f0 + dpp_nb(f0), t1 = f1 + dpp_nb(f1);
        if (hk == 0 && (col & 1) == 0) {
            slab[4 * DD + (col >> 1)] = t0;
            slab[4 * DD + 16 + (col >> 1)] = t1;
        }
    }
    float* tail = slab + 4 * DD + 3 * DPW + 2;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int a = (q & 3) + 8 * (q >> 2) + 4 * hk;
        if (a < r) {
            tail[((col & 1) ? r + a : a) * DPW + (col >> 1)] = g0[q];
            tail[((col & 1) ? r + a : a) * DPW + 16 + (col >> 1)] = g1[q];
        }
    }
    const float sumA = sum64(accA), sumS = sum64(accS);
    // accS = sum_k Re(u^dagger (Q + s_k R^dagger) ybar); its Q part is removed by k_finalize from the reduced Qbar (Dev::abar_fix)
    if (lane == 0) slab[4 * DD + 3 * DPW] = sumA - sumS / A;
}

// ------------------------------------------------------------------------------------------------
// RhoCMPS.sample / rho_evolve_with_sampling / purity (model.py:86-116, 160-167) in the same row-array form: one wavefront per
// path.  The sample needs the expectation BEFORE the update, so a step is  V = U W_R  and  QU = U W_Q  (constant operands:
// their bf16 pieces are split once), then  e = 2 sum U . V  (= Re tr((R + R^dagger) rho), :189-196),  inc = e dt + noise,
// s = inc / A,  Y = (U + QU) + s V,  n = sum Y^2,  U' = rho_k (.) Y / sqrt(n).  Noise is [n_paths][length] (one 64-step chunk
// per lane load), the waveform is written back the same way.  Stash (save): layout 2 rows of (y[n], 0).
// ------------------------------------------------------------------------------------------------
template <bool SAVE, bool F16>
__global__ __launch_bounds__(64 * WAVES, 1) void k_sample_rho_mfma(Dev P, RhoDev W, const float* __restrict__ noise, int n_paths,
                                                                   int length, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float Urow[WAVES][32 * RRLD];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int b = blockIdx.x * WAVES + w;
    if (b >= n_paths) return;
    const int r = W.rank;
    const int col = lane & 31, hk = lane >> 5;
    float* U = &Urow[w][0];
    // pieces of the W forms of R and Q (bf16 x 3, or fp16 x 2 of the forms scaled to 2^15: both operands are constant, and U has
    // unit trace, so every scale is fixed before the first step): lane (col, hk) holds W[16 ks + 8 hk + e][32 t + col], e = 0..7
    unsigned RH[2][4][4], RM[F16 ? 1 : 2][4][4], RL[2][4][4], QH[2][4][4], QM[F16 ? 1 : 2][4][4], QL[2][4][4];
    float sR = 1.f, sQ = 1.f;
    if constexpr (F16) {
        float mr = 0.f, mq = 0.f;
        for (int idx = lane; idx < DPW * DPW; idx += 64) {
            const float2 rr = P.R[idx], qq = P.Q[idx];
            mr = fmaxf(mr, fmaxf(fabsf(rr.x), fabsf(rr.y)));
            mq = fmaxf(mq, fmaxf(fabsf(qq.x), fabsf(qq.y)));
        }
        sR = pow2_below(max64(mr), 15);
        sQ = pow2_below(max64(mq), 15);
    }
    constexpr float sU = 8192.f;
    const float iUR = F16 ? pow2_recip(sU) * pow2_recip(sR) : 1.f, iUQ = F16 ? pow2_recip(sU) * pow2_recip(sQ) : 1.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int n = 32 * t + col, ii = n >> 1, cc = n & 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float wr[8], wq[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int m = 16 * ks + 8 * hk + e, jj = m >> 1, cp = m & 1;
                wr[e] = wform(P.R[ii * DPW + jj], cp, cc);
                wq[e] = wform(P.Q[ii * DPW + jj], cp, cc);
            }
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
                if constexpr (F16) {
                    split2h(wr[2 * e2] * sR, wr[2 * e2 + 1] * sR, RH[t][ks][e2], RL[t][ks][e2]);
                    split2h(wq[2 * e2] * sQ, wq[2 * e2 + 1] * sQ, QH[t][ks][e2], QL[t][ks][e2]);
                } else {
                    split3(wr[2 * e2], wr[2 * e2 + 1], RH[t][ks][e2], RM[t][ks][e2], RL[t][ks][e2]);
                    split3(wq[2 * e2], wq[2 * e2 + 1], QH[t][ks][e2], QM[t][ks][e2], QL[t][ks][e2]);
                }
            }
        }
    }
    for (int a = 0; a < 32; ++a) {                                   // rows a < rank of U = phi_a (model.py:127-136), the rest zero
        float v = 0.f;
        if (a < r) {
            const float2 p = W.phi0[a * DPW + (lane >> 1)];
            v = (lane & 1) ? p.y : p.x;
        }
        U[a * RRLD + lane] = v;
    }
    const float* nrow = noise + (size_t)b * length;
    float* orow = out + (size_t)b * length;
    float2* st = SAVE ? reinterpret_cast<float2*>(W.stash) + (size_t)b * length * r * 64 : nullptr;
    const float A = dev_A(P);
    const float sgn = (col & 1) ? 1.f : -1.f;                        // Im lanes add rho_y * partner, Re lanes subtract it
    float samp = 0.f;
    for (int kbeg = 0; kbeg < length; kbeg += CH) {
        const int cnt = (length - kbeg) < CH ? (length - kbeg) : CH;
        const float nzv = kbeg + lane < length ? nrow[kbeg + lane] : 0.f;
        float outv = 0.f;
        for (int kk = 0; kk < cnt; ++kk) {
            const int k = kbeg + kk;
            const float2 rh0 = P.rho[(size_t)k * DPW + (col >> 1)], rh1 = P.rho[(size_t)k * DPW + 16 + (col >> 1)];
            v16f v0 = {}, v1 = {}, q0 = {}, q1 = {};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const float4 f0 = *reinterpret_cast<const float4*>(U + col * RRLD + 16 * ks + 8 * hk);
                const float4 f1 = *reinterpret_cast<const float4*>(U + col * RRLD + 16 * ks + 8 * hk + 4);
                unsigned AH[4], AM[4], AL[4];
                if constexpr (F16) {
                    split2h(f0.x * sU, f0.y * sU, AH[0], AL[0]);
                    split2h(f0.z * sU, f0.w * sU, AH[1], AL[1]);
                    split2h(f1.x * sU, f1.y * sU, AH[2], AL[2]);
                    split2h(f1.z * sU, f1.w * sU, AH[3], AL[3]);
                    mfma3(v0, AH, AL, RH[0][ks], RL[0][ks]);
                    mfma3(v1, AH, AL, RH[1][ks], RL[1][ks]);
                    mfma3(q0, AH, AL, QH[0][ks], QL[0][ks]);
                    mfma3(q1, AH, AL, QH[1][ks], QL[1][ks]);
                } else {
                    split3(f0.x, f0.y, AH[0], AM[0], AL[0]);
                    split3(f0.z, f0.w, AH[1], AM[1], AL[1]);
                    split3(f1.x, f1.y, AH[2], AM[2], AL[2]);
                    split3(f1.z, f1.w, AH[3], AM[3], AL[3]);
                    mfma6(v0, AH, AM, AL, RH[0][ks], RM[0][ks], RL[0][ks]);
                    mfma6(v1, AH, AM, AL, RH[1][ks], RM[1][ks], RL[1][ks]);
                    mfma6(q0, AH, AM, AL, QH[0][ks], QM[0][ks], QL[0][ks]);
                    mfma6(q1, AH, AM, AL, QH[1][ks], QM[1][ks], QL[1][ks]);
                }
            }
            if constexpr (F16) {
#pragma unroll
                for (int q = 0; q < 16; ++q) { v0[q] *= iUR; v1[q] *= iUR; q0[q] *= iUQ; q1[q] *= iUQ; }
            }
            // C/D layout: column n = 32 t + col, rows (q & 3) + 8 (q >> 2) + 4 hk
            float u0[16], u1[16];
            float acce = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hk;
                u0[q] = U[row * RRLD + col];
                u1[q] = U[row * RRLD + 32 + col];
                acce = fmaf(u0[q], v0[q], acce);
                acce = fmaf(u1[q], v1[q], acce);
            }
            const float e = 2.0f * sum64(acce);                      // Re tr((Rt + Rt^dagger) rho), :189-196
            const float inc = e * P.dt + rdlane(nzv, kk);            // :162
            samp += inc;                                             // :163
            const float s = inc / A;                                 // :164, 175
            float y0[16], y1[16];
            float accn = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                y0[q] = fmaf(s, v0[q], u0[q] + q0[q]);
                y1[q] = fmaf(s, v1[q], u1[q] + q1[q]);
                accn = fmaf(y0[q], y0[q], accn);
                accn = fmaf(y1[q], y1[q], accn);
            }
            const float n = sum64(accn);
            const float sc1 = sqrtf(1.0f / fmaxf(n, 1e-12f));        // :165
            if (SAVE) {
                float2* sbase = st + (size_t)k * r * 64 + lane;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int s0 = (q & 3) + 8 * (q >> 2);
                    const auto ys = __builtin_amdgcn_permlane32_swap(__float_as_uint(y0[q]), __float_as_uint(y1[q]), false, false);
                    if (s0 < r) sbase[(size_t)s0 * 64] = make_float2(__uint_as_float(ys[0]), 0.f);
                    if (s0 + 4 < r) sbase[(size_t)(s0 + 4) * 64] = make_float2(__uint_as_float(ys[1]), 0.f);
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hk;
                const float t0 = sc1 * y0[q], t1 = sc1 * y1[q];
                U[row * RRLD + col] = fmaf(sgn * rh0.y, dpp_nb(t0), rh0.x * t0);
                U[row * RRLD + 32 + col] = fmaf(sgn * rh1.y, dpp_nb(t1), rh1.x * t1);
            }
            outv = lane == kk ? A * samp : outv;                     // :116
        }
        if (kbeg + lane < length) orow[kbeg + lane] = outv;
    }
}

hipError_t launch_fwd_rho_mfma(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, bool f16, bool grad1, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (save && f16 && grad1) hipLaunchKernelGGL((k_fwd_rho_mfma<true, true, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    else if (save && f16) hipLaunchKernelGGL((k_fwd_rho_mfma<true, true, false>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    else if (save && grad1) hipLaunchKernelGGL((k_fwd_rho_mfma<true, false, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    else if (save) hipLaunchKernelGGL((k_fwd_rho_mfma<true, false, false>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    else if (f16) hipLaunchKernelGGL((k_fwd_rho_mfma<false, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    else hipLaunchKernelGGL((k_fwd_rho_mfma<false, false>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    return hipGetLastError();
}

hipError_t launch_sample_rho_mfma(const Dev& P, const RhoDev& W, const float* noise, int n, int length, float* out, bool save,
                                  bool f16, hipStream_t s) {
    const unsigned nb = (unsigned)((n + WAVES - 1) / WAVES);
    if (save && f16) hipLaunchKernelGGL((k_sample_rho_mfma<true, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, noise, n, length, out);
    else if (save) hipLaunchKernelGGL((k_sample_rho_mfma<true, false>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, noise, n, length, out);
    else if (f16) hipLaunchKernelGGL((k_sample_rho_mfma<false, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, noise, n, length, out);
    else hipLaunchKernelGGL((k_sample_rho_mfma<false, false>), dim3(nb), dim3(64 * WAVES), 0, s, P, W, noise, n, length, out);
    return hipGetLastError();
}

hipError_t launch_bwd_rho_mfma(const Dev& P, const RhoDev& W, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(k_bwd_rho_mfma, dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio);
    return hipGetLastError();
}

}  // namespace cmps
