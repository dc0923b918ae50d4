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

// W[m][n] of a complex 32 x 32 matrix M acting on (re, im)-interleaved vectors: (M y)[n = 2 i + c] = sum_m y[m = 2 j + c'] W[m][n],
// W = Mr_ij for c' = c, -Mi_ij for (c', c) = (1, 0), +Mi_ij for (0, 1)
__device__ __forceinline__ float wform(float2 mij, int cp, int cc) { return cp == cc ? mij.x : (cc ? mij.y : -mij.y); }

}  // namespace

template <bool SAVE>
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
    unsigned HH[2][4][4], HM[2][4][4], HL[2][4][4];
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
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) split3(wh[2 * e2], wh[2 * e2 + 1], HH[t][ks][e2], HM[t][ks][e2], HL[t][ks][e2]);
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
    const float A = P.A;
    const float sgn = (col & 1) ? 1.f : -1.f;                  // Im lanes add rho_y * partner, Re lanes subtract it
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
                split3(f0.x, f0.y, AH[0], AM[0], AL[0]);
                split3(f0.z, f0.w, AH[1], AM[1], AL[1]);
                split3(f1.x, f1.y, AH[2], AM[2], AL[2]);
                split3(f1.z, f1.w, AH[3], AM[3], AL[3]);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    unsigned BH[4], BM[4], BL[4];
#pragma unroll
                    for (int e2 = 0; e2 < 4; ++e2)
                        split3(fmaf(s, WR[t][ks][2 * e2], WQ[t][ks][2 * e2]), fmaf(s, WR[t][ks][2 * e2 + 1], WQ[t][ks][2 * e2 + 1]),
                               BH[e2], BM[e2], BL[e2]);
                    mfma6(t ? a1 : a0, AH, AM, AL, BH, BM, BL);
                }
            }
            // ---- Y = U + U W_k in the C/D layout: column n = 32 t + col, rows (r & 3) + 8 (r >> 2) + 4 hk ----
            float y0[16], y1[16];
            float accn = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * hk;
                y0[q] = U[row * RRLD + col] + a0[q];
                y1[q] = U[row * RRLD + 32 + col] + a1[q];
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
                split3(f0.x, f0.y, AH[0], AM[0], AL[0]);
                split3(f0.z, f0.w, AH[1], AM[1], AL[1]);
                split3(f1.x, f1.y, AH[2], AM[2], AL[2]);
                split3(f1.z, f1.w, AH[3], AM[3], AL[3]);
                mfma6(h0, AH, AM, AL, HH[0][ks], HM[0][ks], HL[0][ks]);
                mfma6(h1, AH, AM, AL, HH[1][ks], HM[1][ks], HL[1][ks]);
            }
            float acce = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acce = fmaf(y0[q], h0[q], acce);
                acce = fmaf(y1[q], h1[q], acce);
            }
            const float n = sum64(accn);                         // tr rho', :200
            const float e = sum64(acce);                         // Re tr(x rho'), :195-196
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
}

hipError_t launch_fwd_rho_mfma(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (save)
        hipLaunchKernelGGL(k_fwd_rho_mfma<true>, dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    else
        hipLaunchKernelGGL(k_fwd_rho_mfma<false>, dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    return hipGetLastError();
}

}  // namespace cmps
