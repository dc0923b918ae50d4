// RhoCMPS, the density-matrix scan (model.py:55-203), general D, correctness-first.
//
// The reference carries rho [B, D, D] and applies  rho' = U rho U^dagger,  U = 1 - (dt sigma^2/2) Rt^dagger Rt + s Rt
// (model.py:172-187), loss += -log(1 + Re tr((Rt + Rt^dagger) rho') x / A) (:166, 189-196), rho = rho' / max(tr rho', eps)
// (:198-203).  rho_0 = W^dagger W / tr (:127-132) has rank r = initial_rank (default D), and U rho U^dagger keeps the rank,
// so this file never forms the D x D matrix in the scan: it carries the r columns phi_a of rho = sum_a phi_a phi_a^dagger
// (phi_a(0) = conj(W[a, :]) / sqrt(tr W^dagger W)) through the SAME rotating-frame step as the pure-state kernels,
//
//     y_a = u_a + Q u_a + s R u_a,     e = sum_a y_a^dagger (R + R^dagger) y_a,     n = sum_a |y_a|^2  (= tr rho'),
//     u_a <- rho_k * y_a / sqrt(max(n, 1e-12)),
//
// coupled only through the two scalars e and n: 3 r D^2 complex MACs per step instead of the 4 D^3 of the matrix form.
// One workgroup owns one clip; thread t owns component t of every column; the columns live in LDS.
// Reverse sweep (cotangents g_a of u_a, convention zbar = dL/dRe z + i dL/dIm z), cf. cmps_block.hip:
//     yhb_a = conj(rho_k) g_a;  dot = sum_a Re(yhat_a^dagger yhb_a);  ybar_a = (yhb_a - yhat_a dot)/sqrt(n) + 2 ebar H y_a
//     Rbar += 2 ebar sum_a y_a y_a^dagger + s sum_a ybar_a u_a^dagger;   Qbar += sum_a ybar_a u_a^dagger
//     g_a  = ybar_a + Q ybar_a + s R^dagger ybar_a;   fbar += dt_k sum_a Im(g_a conj(u_a(k+1)))
#include "cmps_internal.h"

namespace cmps {

template <int NT>
__device__ __forceinline__ float rblock_sum(float v, float* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    constexpr int NW = NT / 64;
    if constexpr (NW == 1) return v;
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    __syncthreads();
    return s;
}

// ------------------------------------------------------------------------------------------------
// forward: RhoCMPS._build_loss_rho (model.py:133-144)
// ------------------------------------------------------------------------------------------------
constexpr int RHO_JB = 8;      // matrix rows fetched ahead per block

// body(j, M1[j][tt], M2[j][tt]) for j = 0 .. D-1 in order, the matrix elements (L2 resident: three D x D tables do not fit L1 above
// D = 32) fetched a block of RHO_JB rows ahead of their use: one L2 round trip per block instead of one per row
template <class Body>
__device__ __forceinline__ void rho_jloop(const float2* __restrict__ M1, const float2* __restrict__ M2, int D, int DP, int tt, Body body) {
    float2 n1[RHO_JB], n2[RHO_JB];
#pragma unroll
    for (int jj = 0; jj < RHO_JB; ++jj) {
        const int j = jj < D ? jj : D - 1;
        n1[jj] = M1[j * DP + tt];
        n2[jj] = M2[j * DP + tt];
    }
    for (int j0 = 0; j0 < D; j0 += RHO_JB) {
        float2 c1[RHO_JB], c2[RHO_JB];
#pragma unroll
        for (int jj = 0; jj < RHO_JB; ++jj) { c1[jj] = n1[jj]; c2[jj] = n2[jj]; }
        if (j0 + RHO_JB < D) {
#pragma unroll
            for (int jj = 0; jj < RHO_JB; ++jj) {
                int j = j0 + RHO_JB + jj;
                j = j < D ? j : D - 1;
                n1[jj] = M1[j * DP + tt];
                n2[jj] = M2[j * DP + tt];
            }
        }
        if (j0 + RHO_JB <= D) {
#pragma unroll
            for (int jj = 0; jj < RHO_JB; ++jj) body(j0 + jj, c1[jj], c2[jj]);
        } else {
#pragma unroll
            for (int jj = 0; jj < RHO_JB; ++jj)
                if (j0 + jj < D) body(j0 + jj, c1[jj], c2[jj]);
        }
    }
}

template <int NT, int CW>
__global__ __launch_bounds__(NT) void k_fwd_rho(Dev P, RhoDev W, const float* __restrict__ audio,
                                                float* __restrict__ loss_out, int save, float2* gcols) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, N = P.N, r = W.rank, rD = r * D;
    // the column arrays live in LDS when they fit and in the workspace (RhoDev::cols, L1 / L2 resident) when they do not
    // (rank * D above ~10000 here: the reference's default rank = D at D > 100, model.py:62-65)
    float2* base = gcols ? gcols + (size_t)blockIdx.x * 4 * rD : sh;
    float2* cur = base;
    float2* nxt = base + rD;
    float* red = reinterpret_cast<float*>(gcols ? sh : sh + 2 * rD);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    // the mat-vecs: thread (tt, grp) owns component tt of the columns a0 .. a0 + CW - 1, a0 = CW (grp + NG i) -- one matrix element is
    // loaded once per chunk of CW columns (round 4; before: once per column, with only D of the NT threads at work)
    const int DPT = (NT == 64 || D <= 64) ? 64 : 128, tt = t & (DPT - 1), grp = t / DPT, NG = NT / DPT;
    const bool mact = tt < D;
    const float* xrow = audio + (size_t)b * P.T;
    float2* st = save ? W.stash + (size_t)b * N * r * DP : nullptr;
    if (act)
        for (int a = 0; a < r; ++a) cur[a * D + t] = W.phi0[a * DP + t];
    float loss = 0.f;
    for (int k = 0; k < N; ++k) {
        const float x = xrow[k + 1] - xrow[k];     // model.py:138
        const float s = x / dev_A(P);                   // :175
        __syncthreads();
        if (mact) {
            for (int a0 = grp * CW; a0 < r; a0 += NG * CW) {
                const int na = (r - a0) < CW ? (r - a0) : CW;
                int co[CW];                              // a ragged last chunk repeats its last column: no branch inside the j loop
#pragma unroll
                for (int c = 0; c < CW; ++c) co[c] = (a0 + (c < na ? c : na - 1)) * D;
                float2 v[CW], q[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) v[c] = q[c] = make_float2(0.f, 0.f);
                rho_jloop(P.RT, P.Q, D, DP, tt, [&](int j, float2 m1, float2 m2) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const float2 uj = cur[co[c] + j];
                        v[c] = cfma(m1, uj, v[c]);                // (R u_a)_t
                        q[c] = cfma_conj_a(m2, uj, q[c]);         // (Q u_a)_t, Q Hermitian
                    }
                });
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c < na) {
                        const int a = a0 + c;
                        const float2 u = cur[a * D + tt];
                        const float2 y = make_float2(u.x + q[c].x + s * v[c].x, u.y + q[c].y + s * v[c].y);   // column of U rho U^dagger, :186
                        nxt[a * D + tt] = y;
                        if (save) st[((size_t)k * r + a) * DP + tt] = y;
                    }
            }
        }
        __syncthreads();
        float pe = 0.f, pn = 0.f;
        if (mact) {
            for (int a0 = grp * CW; a0 < r; a0 += NG * CW) {
                const int na = (r - a0) < CW ? (r - a0) : CW;
                int co[CW];                              // a ragged last chunk repeats its last column: no branch inside the j loop
#pragma unroll
                for (int c = 0; c < CW; ++c) co[c] = (a0 + (c < na ? c : na - 1)) * D;
                float2 hy[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) hy[c] = make_float2(0.f, 0.f);
                rho_jloop(P.RT, P.R, D, DP, tt, [&](int j, float2 m1, float2 m2) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const float2 yj = nxt[co[c] + j];
                        hy[c] = cfma(m1, yj, hy[c]);
                        hy[c] = cfma_conj_a(m2, yj, hy[c]);       // ((R + R^dagger) y_a)_t, :193-194
                    }
                });
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c < na) {
                        const float2 y = nxt[(a0 + c) * D + tt];
                        pe += y.x * hy[c].x + y.y * hy[c].y;
                        pn += y.x * y.x + y.y * y.y;
                    }
            }
        }
        const float e = rblock_sum<NT>(pe, red);                  // Re tr(x rho'), :195-196
        const float n = rblock_sum<NT>(pn, red);                  // tr rho', :200
        loss += -logf(1.0f + (e * x) / dev_A(P));                      // :166, 155
        const float sc = sqrtf(1.0f / fmaxf(n, 1e-12f));          // :201 (columns scale with the square root)
        if (act) {
            const float2 rho = P.rho[(size_t)k * DP + t];
            for (int a = 0; a < r; ++a) nxt[a * D + t] = cmul(rho, cscale(sc, nxt[a * D + t]));
        }
        float2* tmp = cur; cur = nxt; nxt = tmp;
    }
    if (t == 0) loss_out[b] = loss;
}

// ------------------------------------------------------------------------------------------------
// reverse sweep
// ------------------------------------------------------------------------------------------------
template <int NT, int EPT, int CW>
__global__ __launch_bounds__(NT) void k_bwd_rho(Dev P, RhoDev W, const float* __restrict__ audio, float2* gcols) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, N = P.N, r = W.rank, rD = r * D;
    float2* base = gcols ? gcols + (size_t)blockIdx.x * 4 * rD : sh;   // LDS, or the workspace when 4 r D complex numbers do not fit
    float2* Y = base;             // y_a of step k
    float2* YB = base + rD;       // yhb_a, then ybar_a
    float2* U = base + 2 * rD;    // H y_a, then u_a(k)
    float2* G = base + 3 * rD;    // cotangent of u_a(k+1)
    float* red = reinterpret_cast<float*>(gcols ? sh : sh + 4 * rD);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    const int DPT = (NT == 64 || D <= 64) ? 64 : 128, tt = t & (DPT - 1), grp = t / DPT, NG = NT / DPT;   // the mat-vecs: as in k_fwd_rho
    const bool mact = tt < D;
    const float* xrow = audio + (size_t)b * P.T;
    const float2* st = W.stash + (size_t)b * N * r * DP;
    const float2 zero = make_float2(0.f, 0.f);
    float2 Rb[EPT], Qb[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) Rb[m] = Qb[m] = zero;
    float facc = 0.f, Abar = 0.f;
    if (act)
        for (int a = 0; a < r; ++a) G[a * D + t] = zero;

    for (int k = N - 1; k >= 0; --k) {
        const float x = xrow[k + 1] - xrow[k];
        const float s = x / dev_A(P);
        const float2 rho = act ? P.rho[(size_t)k * DP + t] : make_float2(1.f, 0.f);
        float pn = 0.f;
        if (act) {
            for (int a = 0; a < r; ++a) {
                const float2 y = st[((size_t)k * r + a) * DP + t];
                Y[a * D + t] = y;
                pn += y.x * y.x + y.y * y.y;
            }
        }
        const float nraw = rblock_sum<NT>(pn, red);
        const float inv = sqrtf(1.0f / fmaxf(nraw, 1e-12f));
        float pd = 0.f;
        if (act) {
            const float dk = P.dtk[k];
            for (int a = 0; a < r; ++a) {
                const float2 yh = cscale(inv, Y[a * D + t]);
                const float2 un = cmul(rho, yh);
                const float2 g = G[a * D + t];
                facc += dk * (g.y * un.x - g.x * un.y);
                const float2 yhb = cmul_conj_a(rho, g);
                pd += yh.x * yhb.x + yh.y * yhb.y;
                YB[a * D + t] = yhb;
            }
        }
        const float dot = rblock_sum<NT>(pd, red);
        __syncthreads();
        float pe = 0.f;
        if (mact) {
            for (int a0 = grp * CW; a0 < r; a0 += NG * CW) {
                const int na = (r - a0) < CW ? (r - a0) : CW;
                int co[CW];                              // a ragged last chunk repeats its last column: no branch inside the j loop
#pragma unroll
                for (int c = 0; c < CW; ++c) co[c] = (a0 + (c < na ? c : na - 1)) * D;
                float2 hy[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) hy[c] = zero;
                rho_jloop(P.RT, P.R, D, DP, tt, [&](int j, float2 m1, float2 m2) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const float2 yj = Y[co[c] + j];
                        hy[c] = cfma(m1, yj, hy[c]);
                        hy[c] = cfma_conj_a(m2, yj, hy[c]);
                    }
                });
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c < na) {
                        const float2 y = Y[(a0 + c) * D + tt];
                        pe += y.x * hy[c].x + y.y * hy[c].y;
                        U[(a0 + c) * D + tt] = hy[c];
                    }
            }
        }
        const float e = rblock_sum<NT>(pe, red);
        const float ex = e * x;
        const float z = ex / dev_A(P);
        const float zbar = -1.0f / (1.0f + z);
        const float ebar = zbar * x / dev_A(P);
        Abar += zbar * (-ex / (dev_A(P) * dev_A(P)));
        const float te = 2.0f * ebar;
        float pnp = 0.f;
        if (act) {
            for (int a = 0; a < r; ++a) {
                const float2 yhb = YB[a * D + t], hy = U[a * D + t];
                const float2 yh = cscale(inv, Y[a * D + t]);
                float2 yb;
                if (nraw > 1e-12f)
                    yb = make_float2((yhb.x - yh.x * dot) * inv, (yhb.y - yh.y * dot) * inv);
                else
                    yb = cscale(inv, yhb);
                yb.x += te * hy.x;
                yb.y += te * hy.y;
                YB[a * D + t] = yb;
                float2 up;
                if (k > 0) {
                    up = st[((size_t)(k - 1) * r + a) * DP + t];
                    pnp += up.x * up.x + up.y * up.y;
                } else {
                    up = W.phi0[a * DP + t];
                }
                U[a * D + t] = up;
            }
        }
        if (k > 0) {
            const float nprev = rblock_sum<NT>(pnp, red);
            const float invp = sqrtf(1.0f / fmaxf(nprev, 1e-12f));
            if (act) {
                const float2 rhop = P.rho[(size_t)(k - 1) * DP + t];
                for (int a = 0; a < r; ++a) U[a * D + t] = cmul(rhop, cscale(invp, U[a * D + t]));
            }
        }
        __syncthreads();
        float ps = 0.f;
        if (mact) {
            for (int a0 = grp * CW; a0 < r; a0 += NG * CW) {
                const int na = (r - a0) < CW ? (r - a0) : CW;
                int co[CW];                              // a ragged last chunk repeats its last column: no branch inside the j loop
#pragma unroll
                for (int c = 0; c < CW; ++c) co[c] = (a0 + (c < na ? c : na - 1)) * D;
                float2 bq[CW], d[CW];
#pragma unroll
                for (int c = 0; c < CW; ++c) bq[c] = d[c] = zero;
                rho_jloop(P.Q, P.R, D, DP, tt, [&](int j, float2 m1, float2 m2) {
#pragma unroll
                    for (int c = 0; c < CW; ++c) {
                        const float2 yj = YB[co[c] + j];
                        bq[c] = cfma_conj_a(m1, yj, bq[c]);      // (Q ybar_a)_t
                        d[c] = cfma_conj_a(m2, yj, d[c]);        // (R^dagger ybar_a)_t
                    }
                });
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    if (c < na) {
                        const int a = a0 + c;
                        const float2 uk = U[a * D + tt], ybt = YB[a * D + tt];
                        ps += d[c].x * uk.x + d[c].y * uk.y;
                        G[a * D + tt] = make_float2(ybt.x + bq[c].x + s * d[c].x, ybt.y + bq[c].y + s * d[c].y);
                    }
            }
        }
        const float sbar = rblock_sum<NT>(ps, red);
        Abar += sbar * (-x / (dev_A(P) * dev_A(P)));
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = t + m * NT;
            if (idx < D * D) {
                const int i = idx / D, j = idx % D;
                float2 o1 = zero, o2 = zero;
                for (int a = 0; a < r; ++a) {
                    const float2 yi = Y[a * D + i], yj = Y[a * D + j], ybi = YB[a * D + i], uj = U[a * D + j];
                    o1.x += yi.x * yj.x + yi.y * yj.y;
                    o1.y += yi.y * yj.x - yi.x * yj.y;
                    o2.x += ybi.x * uj.x + ybi.y * uj.y;
                    o2.y += ybi.y * uj.x - ybi.x * uj.y;
                }
                Rb[m].x += te * o1.x + s * o2.x;
                Rb[m].y += te * o1.y + s * o2.y;
                Qb[m].x += o2.x;
                Qb[m].y += o2.y;
            }
        }
        __syncthreads();
    }
    // slab: the pure-state layout (psi_0 slots zero) followed by the cotangents of the r initial columns
    float* slab = W.slabs + (size_t)b * W.slab_floats;
    const int DD = DP * DP;
    for (int idx = t; idx < (int)W.slab_floats; idx += NT) slab[idx] = 0.f;
    __syncthreads();
#pragma unroll
    for (int m = 0; m < EPT; ++m) {
        const int idx = t + m * NT;
        if (idx < D * D) {
            const int i = idx / D, j = idx % D, o = i * DP + j;
            slab[o] = Rb[m].x;
            slab[DD + o] = Rb[m].y;
            slab[2 * DD + o] = Qb[m].x;
            slab[3 * DD + o] = Qb[m].y;
        }
    }
    if (act) {
        slab[4 * DD + t] = facc;
        float* tail = slab + 4 * DD + 3 * DP + 2;
        for (int a = 0; a < r; ++a) {
            const float2 g = G[a * D + t];
            tail[a * DP + t] = g.x;
            tail[(r + a) * DP + t] = g.y;
        }
    }
    if (t == 0) slab[4 * DD + 3 * DP] = Abar;
}

// tail of the gradient buffer: d phi_re [r][D] | d phi_im [r][D]
__global__ void k_finalize_rho(Dev P, RhoDev W, float* __restrict__ grad_out) {
    const int D = P.D, DP = P.DP, r = W.rank;
    const float* tail = W.sums + 4 * DP * DP + 3 * DP + 2;
    float* out = grad_out + 2 * D * D + 3 * D + 2;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int idx = tid; idx < 2 * r * D; idx += nth) {
        const int a = idx / D, d = idx % D;      // a in [0, 2r): re rows then im rows
        out[idx] = tail[a * DP + d];
    }
}

// ------------------------------------------------------------------------------------------------
// rho_evolve_with_data / rho_evolve_with_sampling / purity (model.py:76-107): lab-frame rho after every
// step from the stashed columns,  rho_k = phases_k (sum_a y_a y_a^dagger) phases_k^dagger / max(n, eps)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_states_rho(Dev P, RhoDev W, int steps, float* __restrict__ rho_out,
                                                    float* __restrict__ purity_out, int direct) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, r = W.rank;
    // direct != 0: the r columns do not fit in LDS and are read from the stash (L1 / L2) every time they are needed
    float2* Y = sh;                                         // [r][D] (direct == 0)
    float2* ph = direct ? sh : sh + r * D;                  // [D]
    float* red = reinterpret_cast<float*>(ph + D);
    const size_t row = blockIdx.x;                          // b * steps + k
    const int k = (int)(row % steps), t = threadIdx.x;
    const float2* st = W.stash + row * r * DP;
    const float* stw = reinterpret_cast<const float*>(W.stash) + row * r * 128;   // wave layout: [rank][64] (y own, H y own)
    auto ldY = [&](int a, int d) {
        if (W.stash_layout == 3) {                          // the wide kernels' rows (cmps_wide.hip): one vector per pair of columns, lane order
            const size_t vp = ((row / steps) * W.vrank + a) >> 1;
            const float* base = W.vstash + ((vp * steps + k) * 2) * (size_t)(4 * DP);
            const int p0 = 64 * (d >> 4) + 8 * ((((d & 15) >> 3) << 2) | (a & 1)) + (d & 7);
            return make_float2(base[p0], base[p0 + 16]);    // component bit = q bit 1 = + 16 lanes
        }
        return W.stash_layout == 1 ? make_float2(stw[(a * 64 + d) * 2], stw[(a * 64 + d + 32) * 2])
             : W.stash_layout == 2 ? make_float2(stw[(a * 64 + 2 * d) * 2], stw[(a * 64 + 2 * d + 1) * 2]) : st[a * DP + d];
    };
    float pn = 0.f;
    for (int idx = t; idx < r * D; idx += 256) {
        const int a = idx / D, d = idx % D;
        const float2 y = ldY(a, d);
        if (!direct) Y[idx] = y;
        pn += y.x * y.x + y.y * y.y;
    }
    if (t < D) {
        const float th = __fmul_rn(P.freqs[t], P.ttab[k]);
        float sn, cs;
        sincosf(th, &sn, &cs);
        ph[t] = make_float2(cs, sn);
    }
    const float n = rblock_sum<256>(pn, red);
    const float inv = 1.0f / fmaxf(n, 1e-12f);
    __syncthreads();
    float pp = 0.f;
    for (int idx = t; idx < D * D; idx += 256) {
        const int i = idx / D, j = idx % D;
        float2 o = make_float2(0.f, 0.f);
        for (int a = 0; a < r; ++a) {
            const float2 yi = direct ? ldY(a, i) : Y[a * D + i], yj = direct ? ldY(a, j) : Y[a * D + j];
            o.x += yi.x * yj.x + yi.y * yj.y;
            o.y += yi.y * yj.x - yi.x * yj.y;
        }
        o = cscale(inv, o);
        pp += o.x * o.x + o.y * o.y;                        // tr rho^2 = sum |rho_ij|^2 (Hermitian), model.py:96
        if (rho_out) {
            const float2 w = cmul(cmul(ph[i], o), make_float2(ph[j].x, -ph[j].y));
            rho_out[(row * D * D + idx) * 2] = w.x;
            rho_out[(row * D * D + idx) * 2 + 1] = w.y;
        }
    }
    const float pur = rblock_sum<256>(pp, red);
    if (purity_out && t == 0) purity_out[row] = pur;
}

// ------------------------------------------------------------------------------------------------
// RhoCMPS._update_ancilla_rho (model.py:172-187) for a general rho [B][D][D]: one workgroup per (clip, row i)
//   U[i][j] = delta_ij + phases_i (Q[i][j] + s R[i][j]) conj(phases_j);   out = (U rho) U^dagger
// ------------------------------------------------------------------------------------------------
__global__ void k_update_ancilla_rho(Dev P, const float* __restrict__ rho_in, const float* __restrict__ signal,
                                     float tt, float* __restrict__ rho_out) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP;
    float2* ph = sh;            // [D]
    float2* Ui = sh + D;        // row i of U
    float2* Ti = sh + 2 * D;    // row i of U rho
    const int b = blockIdx.x / D, i = blockIdx.x % D, t = threadIdx.x;
    const bool act = t < D;
    const float s = signal[b] / dev_A(P);
    if (act) {
        const float th = __fmul_rn(P.freqs[t], tt);
        float sn, cs;
        sincosf(th, &sn, &cs);
        ph[t] = make_float2(cs, sn);
    }
    __syncthreads();
    auto Uel = [&](int a, int c) {
        const float2 q = P.Q[c * DP + a];           // Q[a][c] = conj(Q[c][a])
        const float2 rr = P.RT[c * DP + a];         // R[a][c]
        const float2 m = make_float2(q.x + s * rr.x, -q.y + s * rr.y);
        float2 u = cmul(cmul(ph[a], m), make_float2(ph[c].x, -ph[c].y));
        if (a == c) u.x += 1.0f;
        return u;
    };
    if (act) Ui[t] = Uel(i, t);
    __syncthreads();
    const float* rin = rho_in + (size_t)b * D * D * 2;
    if (act) {
        float2 acc = make_float2(0.f, 0.f);
        for (int j = 0; j < D; ++j)
            acc = cfma(Ui[j], make_float2(rin[(j * D + t) * 2], rin[(j * D + t) * 2 + 1]), acc);
        Ti[t] = acc;
    }
    __syncthreads();
    if (act) {
        float2 acc = make_float2(0.f, 0.f);        // out[i][l=t] = sum_k T[i][k] conj(U[l][k])
        for (int k = 0; k < D; ++k) {
            const float2 u = Uel(t, k);
            acc = cfma(Ti[k], make_float2(u.x, -u.y), acc);
        }
        float* o = rho_out + ((size_t)b * D * D + (size_t)i * D + t) * 2;
        o[0] = acc.x;
        o[1] = acc.y;
    }
}

// ------------------------------------------------------------------------------------------------
// RhoCMPS.sample / rho_evolve_with_sampling / purity (model.py:86-116, 160-167)
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void k_sample_rho(Dev P, RhoDev W, const float* __restrict__ noise, int length,
                                                   float* __restrict__ out, int save, float2* gcols) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, r = W.rank, rD = r * D;
    float2* base = gcols ? gcols + (size_t)blockIdx.x * 4 * rD : sh;
    float2* S = base;             // u_a
    float2* Vb = base + rD;       // R u_a
    float2* Wb = base + 2 * rD;   // u_a + Q u_a
    float* red = reinterpret_cast<float*>(gcols ? sh : sh + 3 * rD);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    float2* st = save ? W.stash + (size_t)b * length * r * DP : nullptr;
    if (act)
        for (int a = 0; a < r; ++a) S[a * D + t] = W.phi0[a * DP + t];
    float samp = 0.f;
    for (int k = 0; k < length; ++k) {
        __syncthreads();
        float pe = 0.f;
        if (act) {
            for (int a = 0; a < r; ++a) {
                const float2* ua = S + a * D;
                float2 v = make_float2(0.f, 0.f), q = make_float2(0.f, 0.f);
                for (int j = 0; j < D; ++j) {
                    const float2 uj = ua[j];
                    v = cfma(P.RT[j * DP + t], uj, v);
                    q = cfma_conj_a(P.Q[j * DP + t], uj, q);
                }
                const float2 u = ua[t];
                pe += u.x * v.x + u.y * v.y;
                Vb[a * D + t] = v;
                Wb[a * D + t] = cadd(u, q);
            }
        }
        const float e = 2.0f * rblock_sum<NT>(pe, red);                            // Re tr((Rt + Rt^dagger) rho), :189-196
        const float inc = e * P.dt + noise[(size_t)b * length + k];                // :162
        samp += inc;                                                               // :163
        const float s = inc / dev_A(P);                                                 // :164, 175
        float pn = 0.f;
        if (act) {
            for (int a = 0; a < r; ++a) {
                const float2 w = Wb[a * D + t], v = Vb[a * D + t];
                const float2 y = make_float2(w.x + s * v.x, w.y + s * v.y);
                Wb[a * D + t] = y;
                pn += y.x * y.x + y.y * y.y;
                if (save) st[((size_t)k * r + a) * DP + t] = y;
            }
        }
        const float n = rblock_sum<NT>(pn, red);
        const float sc = sqrtf(1.0f / fmaxf(n, 1e-12f));                           // :165
        __syncthreads();
        if (act) {
            const float2 rho = P.rho[(size_t)k * DP + t];
            for (int a = 0; a < r; ++a) S[a * D + t] = cmul(rho, cscale(sc, Wb[a * D + t]));
        }
        if (t == 0) out[(size_t)b * length + k] = dev_A(P) * samp;                      // :116
    }
}

// pack the r initial columns into the DP-strided table
__global__ void k_pack_phi(int D, int DP, int r, const float* __restrict__ re, const float* __restrict__ im,
                           float2* __restrict__ phi0) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= r * DP) return;
    const int a = idx / DP, d = idx % DP;
    phi0[idx] = d < D ? make_float2(re[a * D + d], im[a * D + d]) : make_float2(0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
template <typename K>
static hipError_t want_lds(K kernel, size_t shm) {
    if (shm <= 64 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
}

hipError_t launch_pack_phi(const Dev& P, const RhoDev& W, const float* re, const float* im, hipStream_t s) {
    const int n = W.rank * P.DP;
    hipLaunchKernelGGL(k_pack_phi, dim3((n + 255) / 256), dim3(256), 0, s, P.D, P.DP, W.rank, re, im,
                       const_cast<float2*>(W.phi0));
    return hipGetLastError();
}

// beyond RHO_LDS_MAX (cmps_internal.h: rho_cols_spill) the column arrays go to RhoDev::cols (the workspace)
static float2* cols_if_needed(const RhoDev& W, size_t want_lds_bytes, size_t& shm, int blocks) {
    if (want_lds_bytes <= RHO_LDS_MAX) { shm = want_lds_bytes; return nullptr; }
    shm = 1024;                                   // reduction scratch (and the phases of k_states_rho) only
    return (W.cols && blocks <= W.cols_blocks) ? W.cols : reinterpret_cast<float2*>(1);   // 1: "needed but not provided"
}

// columns per mat-vec chunk (accumulators in registers): 8 when every column group has at least 8 columns, else 4
template <int NT, int CW>
static hipError_t run_fwd_rho(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, float2* g, size_t shm, hipStream_t s) {
    hipError_t e = want_lds(k_fwd_rho<NT, CW>, shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_fwd_rho<NT, CW>), dim3(P.B), dim3(NT), shm, s, P, W, audio, loss, save ? 1 : 0, g);
    return hipGetLastError();
}
template <int NT, int EPT, int CW>
static hipError_t run_bwd_rho(const Dev& P, const RhoDev& W, const float* audio, float2* g, size_t shm, hipStream_t s) {
    hipError_t e = want_lds(k_bwd_rho<NT, EPT, CW>, shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_bwd_rho<NT, EPT, CW>), dim3(P.B), dim3(NT), shm, s, P, W, audio, g);
    return hipGetLastError();
}

hipError_t launch_fwd_rho(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, hipStream_t s) {
    size_t shm;
    float2* g = cols_if_needed(W, (size_t)2 * W.rank * P.D * sizeof(float2) + 128, shm, P.B);
    if (g == reinterpret_cast<float2*>(1)) return hipErrorInvalidValue;
    if (P.D <= 32) return W.rank >= 8 ? run_fwd_rho<64, 8>(P, W, audio, loss, save, g, shm, s) : run_fwd_rho<64, 4>(P, W, audio, loss, save, g, shm, s);
    const bool wide = W.rank >= 32;               // four column groups (of 64 threads up to D = 64, of 128 above)
    if (P.D <= 64) return wide ? run_fwd_rho<256, 8>(P, W, audio, loss, save, g, shm, s) : run_fwd_rho<256, 4>(P, W, audio, loss, save, g, shm, s);
    return wide ? run_fwd_rho<512, 8>(P, W, audio, loss, save, g, shm, s) : run_fwd_rho<512, 4>(P, W, audio, loss, save, g, shm, s);
}

hipError_t launch_bwd_rho(const Dev& P, const RhoDev& W, const float* audio, hipStream_t s) {
    size_t shm;
    float2* g = cols_if_needed(W, (size_t)4 * W.rank * P.D * sizeof(float2) + 128, shm, P.B);
    if (g == reinterpret_cast<float2*>(1)) return hipErrorInvalidValue;
    if (P.D <= 32) return W.rank >= 8 ? run_bwd_rho<64, 16, 8>(P, W, audio, g, shm, s) : run_bwd_rho<64, 16, 4>(P, W, audio, g, shm, s);
    if (P.D <= 64) return W.rank >= 32 ? run_bwd_rho<256, 16, 8>(P, W, audio, g, shm, s) : run_bwd_rho<256, 16, 4>(P, W, audio, g, shm, s);
    return W.rank >= 64 ? run_bwd_rho<1024, 16, 8>(P, W, audio, g, shm, s) : run_bwd_rho<1024, 16, 4>(P, W, audio, g, shm, s);   // eight groups
}

hipError_t launch_finalize_rho(const Dev& P, const RhoDev& W, float* grad_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize_rho, dim3(8), dim3(256), 0, s, P, W, grad_out);
    return hipGetLastError();
}

hipError_t launch_states_rho(const Dev& P, const RhoDev& W, int B, int steps, float* rho_out, float* purity_out,
                             hipStream_t s) {
    size_t shm = ((size_t)W.rank * P.D + P.D) * sizeof(float2) + 128;
    const int direct = shm > RHO_LDS_MAX ? 1 : 0;
    if (direct) shm = (size_t)P.D * sizeof(float2) + 128;
    hipError_t e = want_lds(k_states_rho, shm);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_states_rho, dim3((unsigned)((size_t)B * steps)), dim3(256), shm, s, P, W, steps, rho_out, purity_out, direct);
    return hipGetLastError();
}

hipError_t launch_update_ancilla_rho(const Dev& P, const float* rho_in, const float* signal, float t, int B,
                                     float* rho_out, hipStream_t s) {
    const size_t shm = (size_t)3 * P.D * sizeof(float2);
    const int nt = (P.D + 63) / 64 * 64;
    hipLaunchKernelGGL(k_update_ancilla_rho, dim3((unsigned)(B * P.D)), dim3(nt), shm, s, P, rho_in, signal, t, rho_out);
    return hipGetLastError();
}

hipError_t launch_sample_rho(const Dev& P, const RhoDev& W, const float* noise, int n, int length, float* out,
                             bool save, hipStream_t s) {
    size_t shm;
    float2* g = cols_if_needed(W, (size_t)3 * W.rank * P.D * sizeof(float2) + 128, shm, n);
    if (g == reinterpret_cast<float2*>(1)) return hipErrorInvalidValue;
    hipError_t e;
    if (P.D <= 64) {
        if ((e = want_lds(k_sample_rho<64>, shm)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_sample_rho<64>, dim3(n), dim3(64), shm, s, P, W, noise, length, out, save ? 1 : 0, g);
    } else {
        if ((e = want_lds(k_sample_rho<128>, shm)) != hipSuccess) return e;
        hipLaunchKernelGGL(k_sample_rho<128>, dim3(n), dim3(128), shm, s, P, W, noise, length, out, save ? 1 : 0, g);
    }
    return hipGetLastError();
}

}  // namespace cmps
