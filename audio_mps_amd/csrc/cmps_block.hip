// Block-per-clip kernels: the general-D (1..128) form of the scan.  One workgroup owns one clip; the
// D-vector state is broadcast through LDS, R / R^T / Q are read from global memory (cache resident).
// This variant favours clarity over speed; the wave-per-clip kernels in cmps_wave.hip are the hot path
// for D <= 32.  Both compute the same rotating-frame recurrence:
//
//   u_0 = psi_0;  for k = 0..N-1:   s = x_k / A                         model.py:263, 303
//     y = u_k + Q u_k + s R u_k        (Q = -(dt sigma^2/2) R^dagger R)   model.py:306-317
//     e = 2 Re(y^dagger R y);  loss += -log(1 + (e x_k)/A)              model.py:322-325, 294, 279
//     n = max(|y|^2, 1e-12);  u_{k+1} = rho_k * y / sqrt(n)              model.py:331-334 (+ phases :305)
//
// where u_k = psi_k * conj(phases_k) is the reference's `Upsi` and rho_k = phases_k conj(phases_{k+1}).
#include "cmps_internal.h"

namespace cmps {

// Sum over the workgroup, result to every thread; fixed order (deterministic).  Two barriers.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
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
// forward
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void k_fwd_block(Dev P, const float* __restrict__ audio,
                                                  float* __restrict__ loss_out, int save) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, N = P.N;
    float2* su = sh;
    float2* sy = sh + D;
    float* red = reinterpret_cast<float*>(sh + 2 * D);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    const float* xrow = audio + (size_t)b * P.T;
    float2 u = act ? P.psi0[t] : make_float2(0.f, 0.f);
    float loss = 0.f;
    float xprev = xrow[0];
    const float A = dev_A(P);
    for (int k = 0; k < N; ++k) {
        const float xcur = xrow[k + 1];
        const float x = xcur - xprev;              // model.py:263
        xprev = xcur;
        const float s = x / A;                     // model.py:303
        if (act) su[t] = u;
        __syncthreads();
        float2 v = make_float2(0.f, 0.f), q = make_float2(0.f, 0.f);
        if (act) {
            for (int j = 0; j < D; ++j) {
                const float2 uj = su[j];
                v = cfma(P.RT[j * DP + t], uj, v);          // (R u)_t
                q = cfma_conj_a(P.Q[j * DP + t], uj, q);    // (Q u)_t, Q Hermitian
            }
        }
        const float2 y = make_float2(u.x + q.x + s * v.x, u.y + q.y + s * v.y);
        if (act) sy[t] = y;
        __syncthreads();
        float2 r = make_float2(0.f, 0.f);
        if (act)
            for (int j = 0; j < D; ++j) r = cfma(P.RT[j * DP + t], sy[j], r);
        const float pe = act ? (y.x * r.x + y.y * r.y) : 0.f;
        const float pn = act ? (y.x * y.x + y.y * y.y) : 0.f;
        const float e = 2.0f * block_sum<NT>(pe, red);      // model.py:325
        const float n = block_sum<NT>(pn, red);
        const float z = (e * x) / A;                        // model.py:294
        loss += -logf(1.0f + z);                            // model.py:279
        if (save && act) P.stash[((size_t)b * N + k) * DP + t] = y;
        const float inv = 1.0f / sqrtf(fmaxf(n, 1e-12f));   // model.py:332
        if (act) u = cmul(P.rho[(size_t)k * DP + t], cscale(inv, y));
    }
    if (t == 0) loss_out[b] = loss;
}

// ------------------------------------------------------------------------------------------------
// backward (reverse sweep over the stash).  Cotangent convention zbar = dL/dRe z + i dL/dIm z.
//   g = cotangent of u_{k+1};  yhat = y / sqrt(n)
//   fbar   += (t_k - t_{k+1}) * Im(g conj(u_{k+1}))
//   yhb     = conj(rho_k) g ;  ybar = (yhb - yhat Re(yhat^dagger yhb)) / sqrt(n)
//   zbar    = -1 / (1 + z);  ebar = zbar x / A;  Abar += zbar * (-(e x) / A^2)
//   ybar   += 2 ebar (R y + R^dagger y);             Rbar += 2 ebar y y^dagger
//   d = R^dagger ybar;  sbar = Re(d^dagger u_k);      Abar += sbar * (-x / A^2)
//   Qbar   += ybar u_k^dagger;                        Rbar += s ybar u_k^dagger
//   g       = ybar + Q ybar + s d
// ------------------------------------------------------------------------------------------------
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void k_bwd_block(Dev P, const float* __restrict__ audio) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, N = P.N;
    float2* sy = sh;            // y_k
    float2* syb = sh + D;       // ybar
    float2* su = sh + 2 * D;    // u_k
    float* red = reinterpret_cast<float*>(sh + 3 * D);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    const float* xrow = audio + (size_t)b * P.T;
    const float2* st = P.stash + (size_t)b * N * DP;
    const float2 zero = make_float2(0.f, 0.f);

    float2 Rb[EPT], Qb[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) Rb[m] = Qb[m] = zero;
    float facc = 0.f, Abar = 0.f;
    float2 g = zero;
    const float A = dev_A(P);

    // state of step N-1
    float2 y = act ? st[(size_t)(N - 1) * DP + t] : zero;
    float nraw = block_sum<NT>(act ? (y.x * y.x + y.y * y.y) : 0.f, red);
    float inv = 1.0f / sqrtf(fmaxf(nraw, 1e-12f));
    float2 yhat = cscale(inv, y);
    float2 unext = act ? cmul(P.rho[(size_t)(N - 1) * DP + t], yhat) : zero;

    for (int k = N - 1; k >= 0; --k) {
        const float x = xrow[k + 1] - xrow[k];
        const float s = x / A;
        const float2 rho = act ? P.rho[(size_t)k * DP + t] : make_float2(1.f, 0.f);
        if (act) facc += P.dtk[k] * (g.y * unext.x - g.x * unext.y);
        const float2 yhb = cmul_conj_a(rho, g);
        const float dot = block_sum<NT>(act ? (yhat.x * yhb.x + yhat.y * yhb.y) : 0.f, red);
        float2 ybar;
        if (nraw > 1e-12f)
            ybar = make_float2((yhb.x - yhat.x * dot) * inv, (yhb.y - yhat.y * dot) * inv);
        else
            ybar = cscale(inv, yhb);
        if (act) sy[t] = y;
        __syncthreads();
        float2 r = zero, a = zero;
        if (act) {
            for (int j = 0; j < D; ++j) {
                const float2 yj = sy[j];
                r = cfma(P.RT[j * DP + t], yj, r);          // (R y)_t
                a = cfma_conj_a(P.R[j * DP + t], yj, a);    // (R^dagger y)_t
            }
        }
        const float e = 2.0f * block_sum<NT>(act ? (y.x * r.x + y.y * r.y) : 0.f, red);
        const float ex = e * x;
        const float z = ex / A;
        const float zbar = -1.0f / (1.0f + z);
        const float ebar = zbar * x / A;
        Abar += zbar * (-ex / (A * A));
        ybar.x += 2.0f * ebar * (r.x + a.x);
        ybar.y += 2.0f * ebar * (r.y + a.y);
        if (act) syb[t] = ybar;
        // u_k from the previous stash entry (or psi_0)
        float2 yprev = zero, yhatp = zero, uk = zero;
        float nprev = 1.f, invp = 1.f;
        if (k > 0) {
            yprev = act ? st[(size_t)(k - 1) * DP + t] : zero;
            nprev = block_sum<NT>(act ? (yprev.x * yprev.x + yprev.y * yprev.y) : 0.f, red);
            invp = 1.0f / sqrtf(fmaxf(nprev, 1e-12f));
            yhatp = cscale(invp, yprev);
            uk = act ? cmul(P.rho[(size_t)(k - 1) * DP + t], yhatp) : zero;
        } else {
            uk = act ? P.psi0[t] : zero;
        }
        if (act) su[t] = uk;
        __syncthreads();
        float2 bq = zero, d = zero;
        if (act) {
            for (int j = 0; j < D; ++j) {
                const float2 yj = syb[j];
                bq = cfma_conj_a(P.Q[j * DP + t], yj, bq);  // (Q ybar)_t
                d = cfma_conj_a(P.R[j * DP + t], yj, d);    // (R^dagger ybar)_t
            }
        }
        const float sbar = block_sum<NT>(act ? (d.x * uk.x + d.y * uk.y) : 0.f, red);
        Abar += sbar * (-x / (A * A));
        const float te = 2.0f * ebar;
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = t + m * NT;
            if (idx < D * D) {
                const int i = idx / D, j = idx % D;
                const float2 yi = sy[i], yj = sy[j], ybi = syb[i], uj = su[j];
                // yi * conj(yj)
                const float2 o1 = make_float2(yi.x * yj.x + yi.y * yj.y, yi.y * yj.x - yi.x * yj.y);
                // ybi * conj(uj)
                const float2 o2 = make_float2(ybi.x * uj.x + ybi.y * uj.y, ybi.y * uj.x - ybi.x * uj.y);
                Rb[m].x += te * o1.x + s * o2.x;
                Rb[m].y += te * o1.y + s * o2.y;
                Qb[m].x += o2.x;
                Qb[m].y += o2.y;
            }
        }
        __syncthreads();
        g = make_float2(ybar.x + bq.x + s * d.x, ybar.y + bq.y + s * d.y);
        y = yprev; nraw = nprev; inv = invp; yhat = yhatp; unext = uk;
    }
    // per-clip slab: Rbar_re | Rbar_im | Qbar_re | Qbar_im (each DP*DP, row-major) | f | g0_re | g0_im | A | -
    float* slab = P.slabs + (size_t)b * P.slab_floats;
    const int DD = DP * DP;
    for (int idx = t; idx < DD; idx += NT) slab[idx] = slab[DD + idx] = slab[2 * DD + idx] = slab[3 * DD + idx] = 0.f;
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
    if (t < DP) {
        slab[4 * DD + t] = act ? facc : 0.f;
        slab[4 * DD + DP + t] = act ? g.x : 0.f;
        slab[4 * DD + 2 * DP + t] = act ? g.y : 0.f;
    }
    if (t == 0) {
        slab[4 * DD + 3 * DP] = Abar;
        slab[4 * DD + 3 * DP + 1] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// cross-clip reduction (fixed order, double accumulation) and the closing terms
//   Rbar += c_half R (Qbar + Qbar^dagger)       (from Q = c_half R^dagger R)
// grad_out: dR_re [D*D] | dR_im [D*D] | dfreqs [D] | dpsi0_re [D] | dpsi0_im [D] | dA | sum loss
// ------------------------------------------------------------------------------------------------
// two passes, fixed summation order (deterministic): RPART row-groups of clips, then the groups
constexpr int RPART = 32;
__global__ void k_reduce_slabs(Dev P, float* __restrict__ part) {
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= P.slab_floats) return;
    const int g = blockIdx.y;
    if (col == 0 && g == 0 && P.status) P.status[0] = 0u;          // this reverse pass's flag word (k_finalize, next in stream order, sets it)
    const int per = (P.B + RPART - 1) / RPART;
    const int b0 = g * per, b1 = (b0 + per < P.B) ? b0 + per : P.B;
    double acc = 0.0;
    for (int b = b0; b < b1; ++b) acc += (double)P.slabs[(size_t)b * P.slab_floats + col];
    reinterpret_cast<double*>(part)[(size_t)g * P.slab_floats + col] = acc;
}
__global__ void k_reduce_parts(Dev P, const float* __restrict__ part, float* __restrict__ sums) {
    const size_t col = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= P.slab_floats) return;
    double acc = 0.0;
    for (int g = 0; g < RPART; ++g) acc += reinterpret_cast<const double*>(part)[(size_t)g * P.slab_floats + col];
    sums[col] = (float)acc;
}

__global__ void k_finalize(Dev P, const float* __restrict__ sums, const float* __restrict__ loss,
                           float* __restrict__ grad_out) {
    const int D = P.D, DP = P.DP, DD = DP * DP;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int nthreads = gridDim.x * blockDim.x;
    // run-time check of the split-operand arithmetic (cmps_psi_grad_status): an fp16 piece that left its scaled range is Inf, and
    // whatever it fed is Inf / NaN here; flag bit 0 = a gradient sum is non-finite, bit 1 = the loss sum is
    auto flag = [&](unsigned bits) {
        if (P.status) { atomicOr(&P.status[0], bits); atomicOr(&P.status[1], bits); }
    };
    bool bad = false;
    for (int idx = tid; idx < D * D; idx += nthreads) {
        const int i = idx / D, j = idx % D;
        double ar = sums[i * DP + j], ai = sums[DD + i * DP + j];
        double cr = 0.0, ci = 0.0;
        for (int k = 0; k < D; ++k) {
            // H[k][j] = Qbar[k][j] + conj(Qbar[j][k])
            const double hr = (double)sums[2 * DD + k * DP + j] + (double)sums[2 * DD + j * DP + k];
            const double hi = (double)sums[3 * DD + k * DP + j] - (double)sums[3 * DD + j * DP + k];
            const float2 rik = P.R[i * DP + k];
            cr += (double)rik.x * hr - (double)rik.y * hi;
            ci += (double)rik.x * hi + (double)rik.y * hr;
        }
        const float gr = (float)(ar + (double)P.c_half * cr), gi = (float)(ai + (double)P.c_half * ci);
        grad_out[idx] = gr;
        grad_out[D * D + idx] = gi;
        bad |= !(isfinite(gr) && isfinite(gi));
    }
    for (int d = tid; d < D; d += nthreads) {
        const float a = sums[4 * DD + d], b = sums[4 * DD + DP + d], c = sums[4 * DD + 2 * DP + d];
        grad_out[2 * D * D + d] = a;
        grad_out[2 * D * D + D + d] = b;
        grad_out[2 * D * D + 2 * D + d] = c;
        bad |= !(isfinite(a) && isfinite(b) && isfinite(c));
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) flag(1u);
    if (!P.abar_fix) {
        if (tid == 0) {
            grad_out[2 * D * D + 3 * D] = sums[4 * DD + 3 * DP];
            if (!isfinite(sums[4 * DD + 3 * DP])) flag(1u);
        }
    } else if (blockIdx.x == 1 && threadIdx.x < 64) {
        // sum_k Re(u_k^dagger Q ybar_k) = Re sum_ij Q_ij Qbar_ji with Qbar = sum_k ybar_k u_k^dagger (sums[2 DD ...], [3 DD ...])
        double p = 0.0;
        for (int idx = threadIdx.x; idx < D * D; idx += 64) {
            const int a = idx / D, b = idx % D;
            const float2 q = P.Q[a * DP + b];
            p += (double)q.x * (double)sums[2 * DD + b * DP + a] - (double)q.y * (double)sums[3 * DD + b * DP + a];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) p += __shfl_xor(p, off, 64);
        if (threadIdx.x == 0) {
            const float ga = (float)((double)sums[4 * DD + 3 * DP] + p / (double)dev_A(P));
            grad_out[2 * D * D + 3 * D] = ga;
            if (!isfinite(ga)) flag(1u);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {      // sum_b loss_b: one wave, strided partials then a fixed-order tree
        double ls = 0.0;
        for (int b = threadIdx.x; b < P.B; b += 64) ls += (double)loss[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off, 64);
        if (threadIdx.x == 0) {
            grad_out[2 * D * D + 3 * D + 1] = (float)ls;
            if (!isfinite((float)ls)) flag(2u);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// PsiCMPS._update_ancilla_psi (model.py:300-317), lab frame, one step.
// ------------------------------------------------------------------------------------------------
__global__ void k_update_ancilla(Dev P, const float* __restrict__ psi_in,
                                 const float* __restrict__ signal, float t, float* __restrict__ psi_out) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP;
    float2* su = sh;
    float2* sv = sh + D;
    const int b = blockIdx.x, i = threadIdx.x;
    const bool act = i < D;
    const float s = signal[b] / dev_A(P);                                   // :303
    float2 psi = make_float2(0.f, 0.f), ph = make_float2(1.f, 0.f), u = psi;
    if (act) {
        psi = make_float2(psi_in[((size_t)b * D + i) * 2], psi_in[((size_t)b * D + i) * 2 + 1]);
        const float th = __fmul_rn(P.freqs[i], t);                     // :305
        float sn, cs;
        sincosf(th, &sn, &cs);
        ph = make_float2(cs, sn);
        u = cmul_conj_a(ph, psi);                                      // :306
        su[i] = u;
    }
    __syncthreads();
    float2 v = make_float2(0.f, 0.f);
    if (act)
        for (int j = 0; j < D; ++j) v = cfma(P.RT[j * DP + i], su[j], v);      // :309
    if (act) sv[i] = v;
    __syncthreads();
    if (act) {
        float2 w = make_float2(0.f, 0.f);
        for (int j = 0; j < D; ++j) w = cfma_conj_a(P.R[j * DP + i], sv[j], w);  // :310
        const float2 delta = make_float2(P.c_half * w.x + s * v.x, P.c_half * w.y + s * v.y);  // :312-313
        const float2 dp = cmul(ph, delta);                                      // :315
        psi_out[((size_t)b * D + i) * 2] = psi.x + dp.x;                         // :317
        psi_out[((size_t)b * D + i) * 2 + 1] = psi.y + dp.y;
    }
}

// PsiCMPS.psi_evolve_with_data (model.py:231-240): psi_{k+1} = phases_k * y_k / sqrt(max(|y_k|^2, 1e-12))
__global__ void k_states(Dev P, float* __restrict__ psi_out) {
    const int D = P.D, DP = P.DP, N = P.N;
    const size_t row = blockIdx.x;  // b * N + k
    const int k = (int)(row % N), i = threadIdx.x;
    const bool act = i < D;
    float2 y = make_float2(0.f, 0.f);
    if (act) {
        if (P.stash_layout == 1) {   // 16-row wave layout: 64 lanes x (y own, H y own); re on lane i, im on lane i + 32
            const float* r = P.hst + row * 128;
            y = make_float2(r[2 * i], r[2 * (i + 32)]);
        } else if (P.stash_layout == 3) {   // 32-row wave layout: 64 (y[n], (H y)[n]) pairs, n = 2 i + {re, im}
            const float* r = P.hst + row * 128;
            y = make_float2(r[4 * i], r[4 * i + 2]);
        } else if (P.stash_layout == 4) {
            // wide variant (cmps_wide.hip): per pair and step [y | H y][wave][lane], lane = 8 q + i, q = (row half, component, clip)
            const int b = (int)(row / N);
            const float* r = reinterpret_cast<const float*>(P.stash) + (((size_t)(b >> 1) * N + k) * 2) * 4 * DP
                             + (i >> 4) * 64 + (i & 7) + 8 * (((i >> 3) & 1) * 4 + (b & 1));
            y = make_float2(r[0], r[16]);
        } else if (P.stash_layout == 2) {
            // pair variant (cmps_pair.hip): per pair and step [y | H y][clip][re | im][DP] float32
            const int b = (int)(row / N);
            const float* r = reinterpret_cast<const float*>(P.stash) + ((((size_t)(b >> 1) * N + k) * 2 + 0) * 2 + (b & 1)) * 2 * DP;
            y = make_float2(r[i], r[DP + i]);
        } else {
            y = P.stash[row * DP + i];
        }
    }
    float n = act ? (y.x * y.x + y.y * y.y) : 0.f;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    __shared__ float red[2];
    if (blockDim.x > 64) {
        if ((i & 63) == 0) red[i >> 6] = n;
        __syncthreads();
        n = red[0] + red[1];
    }
    if (act) {
        const float inv = 1.0f / sqrtf(fmaxf(n, 1e-12f));
        const float th = __fmul_rn(P.freqs[i], P.ttab[k]);
        float sn, cs;
        sincosf(th, &sn, &cs);
        const float2 o = cmul(make_float2(cs, sn), cscale(inv, y));
        psi_out[(row * D + i) * 2] = o.x;
        psi_out[(row * D + i) * 2 + 1] = o.y;
    }
}


// ------------------------------------------------------------------------------------------------
// PsiCMPS.sample (model.py:242-251, 284-291), general D: one workgroup per sample path.
// ------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(NT) void k_sample_block(Dev P, const float* __restrict__ noise, int length,
                                                     float* __restrict__ out) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP;
    float2* su = sh;
    float* red = reinterpret_cast<float*>(sh + D);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    float2 u = act ? P.psi0[t] : make_float2(0.f, 0.f);
    float samp = 0.f;
    for (int k = 0; k < length; ++k) {
        if (act) su[t] = u;
        __syncthreads();
        float2 v = make_float2(0.f, 0.f), q = make_float2(0.f, 0.f);
        if (act) {
            for (int j = 0; j < D; ++j) {
                const float2 uj = su[j];
                v = cfma(P.RT[j * DP + t], uj, v);
                q = cfma_conj_a(P.Q[j * DP + t], uj, q);
            }
        }
        const float e = 2.0f * block_sum<NT>(act ? (u.x * v.x + u.y * v.y) : 0.f, red);   // model.py:319-325
        const float inc = e * P.dt + noise[(size_t)b * length + k];                        // :286
        samp += inc;                                                                       // :287
        const float s = inc / dev_A(P);                                                         // :288, :303
        const float2 y = make_float2(u.x + q.x + s * v.x, u.y + q.y + s * v.y);
        const float n = block_sum<NT>(act ? (y.x * y.x + y.y * y.y) : 0.f, red);
        const float inv = 1.0f / sqrtf(fmaxf(n, 1e-12f));                                  // :289
        if (act) u = cmul(P.rho[(size_t)k * DP + t], cscale(inv, y));
        if (t == 0) out[(size_t)b * length + k] = dev_A(P) * samp;                              // :251
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static inline int round64(int d) { return (d + 63) / 64 * 64; }

hipError_t launch_fwd_block(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    const size_t shm = (size_t)2 * P.D * sizeof(float2) + 64;
    if (P.D <= 64)
        hipLaunchKernelGGL(k_fwd_block<64>, dim3(P.B), dim3(64), shm, s, P, audio, loss, save ? 1 : 0);
    else
        hipLaunchKernelGGL(k_fwd_block<128>, dim3(P.B), dim3(128), shm, s, P, audio, loss, save ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_bwd_block(const Dev& P, const float* audio, hipStream_t s) {
    const size_t shm = (size_t)3 * P.D * sizeof(float2) + 128;
    if (P.D <= 32)
        hipLaunchKernelGGL((k_bwd_block<64, 16>), dim3(P.B), dim3(64), shm, s, P, audio);
    else if (P.D <= 64)
        hipLaunchKernelGGL((k_bwd_block<256, 16>), dim3(P.B), dim3(256), shm, s, P, audio);
    else
        hipLaunchKernelGGL((k_bwd_block<1024, 16>), dim3(P.B), dim3(1024), shm, s, P, audio);
    return hipGetLastError();
}

hipError_t launch_reduce_only(const Dev& P, hipStream_t s) {
    const unsigned nb = (unsigned)((P.slab_floats + 255) / 256);
    float* part = P.sums + ((P.slab_floats + 63) / 64) * 64;          // RPART x slab doubles behind the sums
    hipLaunchKernelGGL(k_reduce_slabs, dim3(nb, RPART), dim3(256), 0, s, P, part);
    hipLaunchKernelGGL(k_reduce_parts, dim3(nb), dim3(256), 0, s, P, (const float*)part, P.sums);
    return hipGetLastError();
}

// one thread per element of Rbar (a D-term double-precision sum each: 0.41 ms at D = 128 with eight workgroups), at least
// eight workgroups (blocks 0 and 1 also reduce the loss and the Abar correction)
static unsigned finalize_blocks(const Dev& P) {
    const unsigned nb = (unsigned)((P.D * P.D + 255) / 256);
    return nb < 8 ? 8 : nb;
}

hipError_t launch_reduce_finalize(const Dev& P, const float* loss, float* grad_out, hipStream_t s) {
    const hipError_t e = launch_reduce_only(P, s);     // its hipGetLastError() has consumed the launch status
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_finalize, dim3(finalize_blocks(P)), dim3(256), 0, s, P, (const float*)P.sums, loss, grad_out);
    return hipGetLastError();
}

hipError_t launch_finalize_only(const Dev& P, const float* loss, float* grad_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(finalize_blocks(P)), dim3(256), 0, s, P, (const float*)P.sums, loss, grad_out);
    return hipGetLastError();
}

hipError_t launch_update_ancilla(const Dev& P, const float* psi_in, const float* signal, float t,
                                 int B, float* psi_out, hipStream_t s) {
    const size_t shm = (size_t)2 * P.D * sizeof(float2);
    hipLaunchKernelGGL(k_update_ancilla, dim3(B), dim3(round64(P.D)), shm, s, P, psi_in, signal, t, psi_out);
    return hipGetLastError();
}

hipError_t launch_sample_block(const Dev& P, const float* noise, int n, int length, float* out, hipStream_t s) {
    const size_t shm = (size_t)P.D * sizeof(float2) + 64;
    if (P.D <= 64)
        hipLaunchKernelGGL(k_sample_block<64>, dim3(n), dim3(64), shm, s, P, noise, length, out);
    else
        hipLaunchKernelGGL(k_sample_block<128>, dim3(n), dim3(128), shm, s, P, noise, length, out);
    return hipGetLastError();
}

hipError_t launch_states(const Dev& P, int B, float* psi_out, hipStream_t s) {
    hipLaunchKernelGGL(k_states, dim3((unsigned)((size_t)B * P.N)), dim3(round64(P.D)), 0, s, P, psi_out);
    return hipGetLastError();
}

}  // namespace cmps
