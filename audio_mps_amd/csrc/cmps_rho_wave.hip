// RhoCMPS, wave-per-clip kernels for D <= 32 (any rank <= 32): the density matrix as its `rank` columns (see cmps_rho.hip
// for the arithmetic and the reference lines), every column advanced by the same register-resident mat-vec machinery as
// the pure-state wave kernels (cmps_wave_util.h): lane (i, h) keeps 16 columns of row i of R, Q and R + R^dagger, a
// column vector is broadcast through LDS, the two scalars that couple the columns (e = sum_a y_a^dagger H y_a and
// n = sum_a |y_a|^2 = tr rho') are one wave reduction each per step, whatever the rank.  The column states live in LDS
// (one float per lane and column, split layout).  Straight-line float32 code without the latency tricks of the hot path:
// this row is measured in profiles/r1_next_rows.json, not in bench.py.
//
// Reverse sweep per step: pass 1 over the columns builds yhb_a = conj(rho_k) g_a and dot = sum_a Re(yhat_a^dagger yhb_a);
// pass 2 builds ybar_a, broadcasts it, applies Q and R^dagger, updates g_a and feeds the three rank-1 gradient terms to
// six exact float32 MFMAs per column (K = {re, im}: the split layout is the operand layout, as in k_bwd_wave).
#include "cmps_wave_util.h"

namespace cmps {

namespace {

constexpr int RMAX = 32;      // largest rank of this variant

__device__ __forceinline__ void lds_wait_all8(v4f (&o)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]) : : "memory");
}
// broadcast a split-layout vector through LDS and return this lane's 16 complex entries
__device__ __forceinline__ void bcast(unsigned wr, unsigned rd, float mine, v4f (&q)[8]) {
    bcast_issue(wr, rd, mine, q);
    lds_wait_all8(q);
}

}  // namespace

// per-chunk scalar stash of the rho path: [B][NC][2][64] floats: tr rho'_k and e_k, one step per lane
template <bool SAVE>
__global__ __launch_bounds__(64 * WAVES, 1) void k_fwd_rho_wave(Dev P, RhoDev W, const float* __restrict__ audio,
                                                                float* __restrict__ loss_out) {
    __shared__ __attribute__((aligned(16))) float4 stR[WAVES][CH * 16];
    __shared__ __attribute__((aligned(16))) float2 bcU[WAVES][DPW];
    __shared__ float S[WAVES][RMAX][64];        // column states u_a (split layout)
    __shared__ float Yb[WAVES][RMAX][64];       // y_a of the current step
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH, r = W.rank;
    v2f MR[16], MQ[16], MH[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        MR[m] = ld2(&P.R[i * DPW + 16 * h + m]);
        const v2f rt = ld2(&P.RT[i * DPW + 16 * h + m]);
        MH[m] = mk2(MR[m].x + rt.x, MR[m].y - rt.y);
        MQ[m] = ld2(&P.Q[i * DPW + 16 * h + m]);
    }
    const unsigned aUw = lds_addr(&bcU[w][0]) + i * 8 + h * 4, aUr = lds_addr(&bcU[w][0]) + h * 128;
    const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
    const float* xrow = audio + (size_t)b * T;
    float2* st = SAVE ? reinterpret_cast<float2*>(W.stash) + (size_t)b * N * r * 64 + lane : nullptr;
    float* sc = SAVE ? W.scal + (size_t)b * NC * 128 : nullptr;
    const float A = dev_A(P);
    for (int a = 0; a < r; ++a) {
        const float2 p = W.phi0[a * DPW + i];
        S[w][a][lane] = hb ? p.y : p.x;
    }
    float loss = 0.f;
    v4f sr[16], q[8];
    for (int c = 0; c < NC; ++c) {
        const int kbeg = c * CH;
        const int cnt = (N - kbeg) < CH ? (N - kbeg) : CH;
        stage_load<16>(rho4, kbeg, P.N, lane, sr);
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f, x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        const float incv = x1 - x0;                              // model.py:138
        const float sv = incv / A;                               // :175
        stage_commit<16>(stR[w], lane, sr);
        float nvec = 1.f, evec = 0.f;
        for (int kk = 0; kk < cnt; ++kk) {
            const float s = rdlane(sv, kk);
            float accn = 0.f, acce = 0.f;
            for (int a = 0; a < r; ++a) {
                const float u = S[w][a][lane];
                bcast(aUw, aUr, u, q);
                v2f av, aq;
                mv2_lo(MR, MQ, q, av, aq);
                mv2_hi(MR, MQ, q, av, aq);
                const v2f wp = aq + s * av;
                const float y = u + swapadd(wp.x, wp.y);         // column of U rho U^dagger, :186
                Yb[w][a][lane] = y;
                accn += y * y;
                bcast(aUw, aUr, y, q);
                const v2f ah = mv1(MH, q);
                const float hs = swapadd(ah.x, ah.y);            // ((Rt + Rt^dagger) y_a), :193-194
                acce += y * hs;
                if (SAVE) st[((size_t)(kbeg + kk) * r + a) * 64] = make_float2(y, hs);
            }
            const float n = sum64(accn);                         // tr rho', :200
            const float e = sum64(acce);                         // Re tr(x rho'), :195-196
            nvec = lane == kk ? n : nvec;
            evec = lane == kk ? e : evec;
            const float sc1 = sqrtf(1.0f / fmaxf(n, 1e-12f));    // :201 (columns scale with the square root)
            const float2 rh = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(&stR[w][0]) + kk * 256 + i * 8);
            const v2f rho = mk2(rh.x, rh.y);
            for (int a = 0; a < r; ++a) {
                const float y = Yb[w][a][lane];
                const float yo = osig_of(y, hb);
                const v2f un = cmul2(sc1 * mk2(y, yo), rho);
                S[w][a][lane] = un.x;
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

__global__ __launch_bounds__(64 * WAVES, 1) void k_bwd_rho_wave(Dev P, RhoDev W, const float* __restrict__ audio) {
    __shared__ __attribute__((aligned(16))) float4 stR[WAVES][(CH + 1) * 16];   // rho rows of the chunk + the one below it
    __shared__ __attribute__((aligned(16))) float2 bcB[WAVES][DPW];
    __shared__ float G[WAVES][RMAX][64];        // cotangents g_a of u_a(k+1) (split layout)
    __shared__ float Tb[WAVES][RMAX][64];       // yhb_a of the current step
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH, r = W.rank;
    v2f MD[16], MQ[16];                                          // R^dagger and Q rows
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const v2f rt = ld2(&P.RT[i * DPW + 16 * h + m]);          // R[16h+m][i]
        MD[m] = mk2(rt.x, -rt.y);                                // (R^dagger)[i][16h+m]
        MQ[m] = ld2(&P.Q[i * DPW + 16 * h + m]);
    }
    const unsigned aBw = lds_addr(&bcB[w][0]) + i * 8 + h * 4, aBr = lds_addr(&bcB[w][0]) + h * 128;
    const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
    const float* xrow = audio + (size_t)b * T;
    // stash rows of 64 pairs (y, H y): in lane order (layout 1, k_fwd_rho_wave) or per real component n = 2 i + {re, im}
    // (layout 2, k_fwd_rho_mfma)
    const float2* st = reinterpret_cast<const float2*>(W.stash) + (size_t)b * N * r * 64 + (W.stash_layout == 2 ? 2 * i + h : lane);
    const float* sc = W.scal + (size_t)b * NC * 128;
    const float A = dev_A(P);
    for (int a = 0; a < r; ++a) G[w][a][lane] = 0.f;
    v16f Rre = {}, Rim = {}, Qre = {}, Qim = {};
    float facc = 0.f, accA = 0.f, accS = 0.f;
    v4f sr[16], q[8];
    for (int c = NC - 1; c >= 0; --c) {
        const int kbeg = c * CH;
        const int cnt = (N - kbeg) < CH ? (N - kbeg) : CH;
        // rho rows kbeg-1 .. kbeg+63 (row 0 of the buffer is the row below the chunk: u_k of the chunk's first step needs it)
        stage_load<16>(rho4, kbeg, P.N, lane, sr);
        stage_commit<16>(&stR[w][16], lane, sr);
        if (lane < 16) stR[w][lane] = rho4[(size_t)(kbeg > 0 ? kbeg - 1 : 0) * 16 + lane];
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
            const float dtk = rdlane(dtv, kk), xk = rdlane(incv, kk);
            const float invp = kk > 0 ? rdlane(invv, kk - 1) : invbelow;
            const float2 rh = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(&stR[w][16]) + kk * 256 + i * 8);
            const float2 rp = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(&stR[w][16]) + (kk - 1) * 256 + i * 8);
            const v2f rho = mk2(rh.x, rh.y), rhop = mk2(rp.x, rp.y);
            // ---- pass 1: yhb_a and the joint projection ----
            float accd = 0.f;
            for (int a = 0; a < r; ++a) {
                const float2 row = st[((size_t)k * r + a) * 64];
                const float yh = row.x * inv;
                const float g = G[w][a][lane];
                const float go = osig_of(g, hb);
                const float yho = osig_of(yh, hb);
                const v2f un = cmul2(mk2(yh, yho), rho);                   // u_a(k+1)
                facc += dtk * (go * un.x);                               // Im(g conj(u)) summed over both halves later
                const v2f yhb = cmul2_conj_b(mk2(g, go), rho);            // conj(rho_k) g_a
                accd += yh * yhb.x;
                Tb[w][a][lane] = yhb.x;
            }
            const float dot = sum64(accd);
            // ---- pass 2 ----
            for (int a = 0; a < r; ++a) {
                const float2 row = st[((size_t)k * r + a) * 64];
                const float y = row.x, hy = row.y;
                const float yh = y * inv;
                const float yhb = Tb[w][a][lane];
                const float ybar = (yhb - ok * yh * dot) * inv + te * hy;
                bcast(aBw, aBr, ybar, q);
                v2f ad, aq;
                mv2_lo(MD, MQ, q, ad, aq);
                mv2_hi(MD, MQ, q, ad, aq);
                const float d = swapadd(ad.x, ad.y), bq = swapadd(aq.x, aq.y);
                // u_a(k) = rho_{k-1} y_a(k-1) / sqrt(tr)  (the initial column at k = 0)
                float uk, uko;
                if (k > 0) {
                    const float yp = st[((size_t)(k - 1) * r + a) * 64].x * invp;
                    const v2f t = cmul2(mk2(yp, osig_of(yp, hb)), rhop);
                    uk = t.x;
                    uko = t.y;
                } else {
                    const float2 p = W.phi0[a * DPW + i];
                    uk = hb ? p.y : p.x;
                    uko = hb ? -p.x : p.y;
                }
                accS += d * uk * xk;
                G[w][a][lane] = ybar + bq + s * d;
                // rank-1 terms, exact float32 MFMAs (K = {re, im}); the imaginary parts accumulate with the opposite sign
                const float yo = osig_of(y, hb);
                const float a1 = te * y, a2 = s * ybar;
                Rre = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, y, Rre, 0, 0, 0);
                Rim = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, yo, Rim, 0, 0, 0);
                Qre = __builtin_amdgcn_mfma_f32_32x32x2f32(ybar, uk, Qre, 0, 0, 0);
                Qim = __builtin_amdgcn_mfma_f32_32x32x2f32(ybar, uko, Qim, 0, 0, 0);
                Rre = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, uk, Rre, 0, 0, 0);
                Rim = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, uko, Rim, 0, 0, 0);
            }
        }
    }
    // ---- per-clip slab (layout of k_bwd_rho: the pure-state slab followed by the column cotangents) ----
    float* slab = W.slabs + (size_t)b * W.slab_floats;
    constexpr int DD = DPW * DPW;
    for (int idx = lane; idx < (int)W.slab_floats; idx += 64) slab[idx] = 0.f;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int row = (rr & 3) + 8 * (rr >> 2) + 4 * h;        // C/D layout of the 32x32 MFMA: column = lane & 31
        const int o = row * DPW + i;
        slab[o] = Rre[rr];
        slab[DD + o] = -Rim[rr];
        slab[2 * DD + o] = Qre[rr];
        slab[3 * DD + o] = -Qim[rr];
    }
    const float ftot = swapadd(facc, facc);
    if (!hb) slab[4 * DD + i] = ftot;
    float* tail = slab + 4 * DD + 3 * DPW + 2;
    for (int a = 0; a < r; ++a) tail[(hb ? r + a : a) * DPW + i] = G[w][a][lane];
    const float sumA = sum64(accA), sumS = sum64(accS);
    if (lane == 0) slab[4 * DD + 3 * DPW] = sumA - sumS / (A * A);
}

hipError_t launch_fwd_rho_wave(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (save)
        hipLaunchKernelGGL(k_fwd_rho_wave<true>, dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    else
        hipLaunchKernelGGL(k_fwd_rho_wave<false>, dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio, loss);
    return hipGetLastError();
}

hipError_t launch_bwd_rho_wave(const Dev& P, const RhoDev& W, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(k_bwd_rho_wave, dim3(nb), dim3(64 * WAVES), 0, s, P, W, audio);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Round 5: the reverse sweep of the row-array forward (k_fwd_rho_mfma, stash layout 2) as the PURE-STATE wave reverse scan on virtual
// clips.  Given the clip's per-step scalars (tr rho'_k, e_k: RhoDev::scal) the cotangents of the columns do not couple -- the
// normalisation adjoint's radial part is te_{k+1} e_{k+1} for every column (Euler, as in k_bwd_wave) -- so column a of clip b is clip
// b rank + a of k_bwd_wave (cmps_wave.hip; Dev::phi0 switches the indexing), whose cost is linear in the rank and whose rank-1 sums
// run in the CMPS_OPT_RANK1 arithmetic (fp16 x 2 by default).  k_bwd_rho_mfma (bf16 x 3, the same cost at every rank) stays behind
// CMPS_VARIANT_WAVE32 ... it also needs the forward's part of Rbar (RhoDev::p1), which this path ignores: k_bwd_wave forms all three sums.
// ------------------------------------------------------------------------------------------------
__global__ void k_rho_phi_from_slabs(const float* __restrict__ slabs, size_t slab_floats, int B, int rank, int D, int DP,
                                     float* __restrict__ grad_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int DD = DP * DP, G = 2 * D * D + 3 * D + 2;
    if (idx < 2 * D) grad_out[2 * D * D + D + idx] = 0.f;           // the pure-state layout's psi_0 entries: unused here (include/cmps.h)
    if (idx >= rank * D) return;
    const int a = idx / D, d = idx % D;
    double sr = 0.0, si = 0.0;
    for (int b = 0; b < B; ++b) {
        const float* sl = slabs + ((size_t)b * rank + a) * slab_floats + 4 * DD;
        sr += (double)sl[DP + d];
        si += (double)sl[2 * DP + d];
    }
    grad_out[G + idx] = (float)sr;
    grad_out[G + rank * D + idx] = (float)si;
}

hipError_t launch_bwd_rho_virtual_wave(const Dev& P, const RhoDev& W, const float* audio, const float* loss, float* grad_out, int rank1_mode,
                                       int bwd_waves, hipStream_t s) {
    Dev V = P;
    V.B = P.B * W.rank;
    V.hst = reinterpret_cast<float*>(W.stash);
    V.stash = W.stash;
    V.stash_layout = 3;
    V.scal = W.scal;
    V.slabs = W.wslabs;
    V.sums = W.wsums;
    V.slab_floats = (size_t)4 * P.DP * P.DP + 3 * P.DP + 2;
    V.phi0 = W.phi0;
    V.phi_rank = W.rank;
    V.gphi = nullptr;
    V.status = nullptr;
    hipError_t e;
    const bool two = bwd_waves == 2 && rank1_mode == 3 && P.f16_shift == 0;      // 3 = CMPS_RANK1_F16X2: as cmps_psi_loss_bwd chooses
    { KScope ks(two ? "k_bwd_wave2w" : "k_bwd_wave", s); e = two ? launch_bwd_wave2w(V, audio, s) : launch_bwd_wave(V, audio, rank1_mode, s); }
    if (e != hipSuccess) return e;
    KScope ks("reduce + finalize", s);
    e = launch_reduce_only(V, s);
    if (e != hipSuccess) return e;
    Dev F = V;
    F.B = P.B;                                                   // the loss sum runs over the real clips
    F.abar_fix = 1;
    e = launch_finalize_only(F, loss, grad_out, s);
    if (e != hipSuccess) return e;
    const int n = W.rank * P.D > 2 * P.D ? W.rank * P.D : 2 * P.D;
    hipLaunchKernelGGL(k_rho_phi_from_slabs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)W.wslabs, V.slab_floats, P.B,
                       W.rank, P.D, P.DP, grad_out);
    return hipGetLastError();
}

}  // namespace cmps
