// Legacy `AudioMPS` arithmetic: the previous generation of the model, whose
// class body is gone from model.py but whose training graph survives in logging/graph.pbtxt.
//   psi_0 = e_0;  for k = 0..N-1:   x = data[b,k+1] - data[b,k]
//     e    = 2 Re(psi^dagger R psi)                         (graph.pbtxt:11857-12661, on the normalised pre-update psi)
//     loss += (x - e)^2 / 2                                 (:12685-12819)
//     psi' = psi + Q psi + dt x (R psi),  Q = dt (-i H_s - R^T R / 2)   (:12982-14323; Q is built on the host)
//     psi  = psi' rsqrt(max(|psi'|^2, 1e-12))               (:14350-14594)
// Correctness-first general-D form (one workgroup per clip, matrices read through the caches), like cmps_block.hip.
// Reverse sweep (zbar = dL/dRe z + i dL/dIm z; g = cotangent of the normalised psi_{k+1}):
//   ybar = (g - yhat Re(yhat^dagger g)) / sqrt(n);  ebar = e - x
//   vbar = dt x ybar + 2 ebar psi;   Qbar += ybar psi^dagger;   Rbar_c += vbar psi^dagger
//   g    = ybar + Q^dagger ybar + 2 ebar R psi + R^dagger vbar
#include "cmps_internal.h"

namespace cmps {

template <int NT>
__device__ __forceinline__ float lblock_sum(float v, float* red) {
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

// body(j, M1[j][t], M2[j][t]) for j = 0 .. D-1 in order, the matrix elements (L2 resident above D = 32) fetched a block of rows ahead
// of their use (round 4, as rho_jloop in cmps_rho.hip): one L2 round trip per block of 8 rows instead of one per row
template <class Body>
__device__ __forceinline__ void leg_jloop(const float2* __restrict__ M1, const float2* __restrict__ M2, int D, int DP, int t, Body body) {
    constexpr int JB = 8;
    float2 n1[JB], n2[JB];
#pragma unroll
    for (int jj = 0; jj < JB; ++jj) {
        const int j = jj < D ? jj : D - 1;
        n1[jj] = M1[j * DP + t];
        n2[jj] = M2[j * DP + t];
    }
    for (int j0 = 0; j0 < D; j0 += JB) {
        float2 c1[JB], c2[JB];
#pragma unroll
        for (int jj = 0; jj < JB; ++jj) { c1[jj] = n1[jj]; c2[jj] = n2[jj]; }
        if (j0 + JB < D) {
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) {
                int j = j0 + JB + jj;
                j = j < D ? j : D - 1;
                n1[jj] = M1[j * DP + t];
                n2[jj] = M2[j * DP + t];
            }
        }
        if (j0 + JB <= D) {
#pragma unroll
            for (int jj = 0; jj < JB; ++jj) body(j0 + jj, c1[jj], c2[jj]);
        } else {
#pragma unroll
            for (int jj = 0; jj < JB; ++jj)
                if (j0 + jj < D) body(j0 + jj, c1[jj], c2[jj]);
        }
    }
}

template <int NT>
__global__ __launch_bounds__(NT) void k_fwd_legacy(Dev P, const float* __restrict__ audio,
                                                   float* __restrict__ loss_out, int save) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, N = P.N;
    float2* su = sh;
    float* red = reinterpret_cast<float*>(sh + D);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    const float* xrow = audio + (size_t)b * P.T;
    float2 psi = make_float2(t == 0 ? 1.f : 0.f, 0.f);      // one_hot(0, D)
    float loss = 0.f;
    for (int k = 0; k < N; ++k) {
        const float x = xrow[k + 1] - xrow[k];
        const float c = P.dt * x;
        if (save && act) P.stash[((size_t)b * N + k) * DP + t] = psi;
        if (act) su[t] = psi;
        __syncthreads();
        float2 v = make_float2(0.f, 0.f), q = make_float2(0.f, 0.f);
        if (act) {
            leg_jloop(P.RT, P.QT, D, DP, t, [&](int j, float2 m1, float2 m2) {
                const float2 pj = su[j];
                v = cfma(m1, pj, v);
                q = cfma(m2, pj, q);
            });
        }
        const float e = 2.0f * lblock_sum<NT>(act ? (psi.x * v.x + psi.y * v.y) : 0.f, red);
        const float d = x - e;
        loss += d * d / 2.0f;
        const float2 y = make_float2(psi.x + q.x + c * v.x, psi.y + q.y + c * v.y);
        const float n = lblock_sum<NT>(act ? (y.x * y.x + y.y * y.y) : 0.f, red);
        const float inv = 1.0f / sqrtf(fmaxf(n, 1e-12f));
        psi = act ? cscale(inv, y) : make_float2(0.f, 0.f);
        __syncthreads();
    }
    if (t == 0) loss_out[b] = loss;
}

template <int NT, int EPT>
__global__ __launch_bounds__(NT) void k_bwd_legacy(Dev P, const float* __restrict__ audio) {
    extern __shared__ float2 sh[];
    const int D = P.D, DP = P.DP, N = P.N;
    float2* sp = sh;            // psi_k
    float2* syb = sh + D;       // ybar
    float2* svb = sh + 2 * D;   // vbar
    float* red = reinterpret_cast<float*>(sh + 3 * D);
    const int b = blockIdx.x, t = threadIdx.x;
    const bool act = t < D;
    const float* xrow = audio + (size_t)b * P.T;
    const float2* st = P.stash + (size_t)b * N * DP;
    const float2 zero = make_float2(0.f, 0.f);
    float2 Rb[EPT], Qb[EPT];
#pragma unroll
    for (int m = 0; m < EPT; ++m) Rb[m] = Qb[m] = zero;
    float2 g = zero;
    for (int k = N - 1; k >= 0; --k) {
        const float x = xrow[k + 1] - xrow[k];
        const float c = P.dt * x;
        const float2 p = act ? st[(size_t)k * DP + t] : zero;
        if (act) sp[t] = p;
        __syncthreads();
        float2 v = zero, q = zero;
        if (act) {
            leg_jloop(P.RT, P.QT, D, DP, t, [&](int j, float2 m1, float2 m2) {
                const float2 pj = sp[j];
                v = cfma(m1, pj, v);
                q = cfma(m2, pj, q);
            });
        }
        const float e = 2.0f * lblock_sum<NT>(act ? (p.x * v.x + p.y * v.y) : 0.f, red);
        const float2 y = make_float2(p.x + q.x + c * v.x, p.y + q.y + c * v.y);
        const float n = lblock_sum<NT>(act ? (y.x * y.x + y.y * y.y) : 0.f, red);
        const float inv = 1.0f / sqrtf(fmaxf(n, 1e-12f));
        const float2 yhat = cscale(inv, y);
        const float dot = lblock_sum<NT>(act ? (yhat.x * g.x + yhat.y * g.y) : 0.f, red);
        float2 ybar;
        if (n > 1e-12f)
            ybar = make_float2((g.x - yhat.x * dot) * inv, (g.y - yhat.y * dot) * inv);
        else
            ybar = cscale(inv, g);
        const float te = 2.0f * (e - x);
        const float2 vbar = make_float2(c * ybar.x + te * p.x, c * ybar.y + te * p.y);
        if (act) { syb[t] = ybar; svb[t] = vbar; }
        __syncthreads();
        float2 a = zero, r = zero;
        if (act) {
            leg_jloop(P.Q, P.R, D, DP, t, [&](int j, float2 m1, float2 m2) {
                a = cfma_conj_a(m1, syb[j], a);     // (Q^dagger ybar)_t
                r = cfma_conj_a(m2, svb[j], r);     // (R^dagger vbar)_t
            });
        }
#pragma unroll
        for (int m = 0; m < EPT; ++m) {
            const int idx = t + m * NT;
            if (idx < D * D) {
                const int i = idx / D, j = idx % D;
                const float2 ybi = syb[i], vbi = svb[i], pj = sp[j];
                Qb[m].x += ybi.x * pj.x + ybi.y * pj.y;
                Qb[m].y += ybi.y * pj.x - ybi.x * pj.y;
                Rb[m].x += vbi.x * pj.x + vbi.y * pj.y;
                Rb[m].y += vbi.y * pj.x - vbi.x * pj.y;
            }
        }
        __syncthreads();
        g = make_float2(ybar.x + a.x + te * v.x + r.x, ybar.y + a.y + te * v.y + r.y);
    }
    float* slab = P.slabs + (size_t)b * P.slab_floats;
    const int DD = DP * DP;
    for (int idx = t; idx < (int)P.slab_floats; idx += NT) slab[idx] = 0.f;
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
}

// pack R (real), Q (complex) and their transposes into the DP-strided tables
__global__ void k_pack_legacy(int D, int DP, const float* __restrict__ Rr, const float* __restrict__ Qre,
                              const float* __restrict__ Qim, float2* __restrict__ R, float2* __restrict__ RT,
                              float2* __restrict__ Q, float2* __restrict__ QT) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= DP * DP) return;
    const int i = idx / DP, j = idx % DP;
    const bool in = i < D && j < D;
    R[idx] = in ? make_float2(Rr[i * D + j], 0.f) : make_float2(0.f, 0.f);
    RT[idx] = in ? make_float2(Rr[j * D + i], 0.f) : make_float2(0.f, 0.f);
    Q[idx] = in ? make_float2(Qre[i * D + j], Qim[i * D + j]) : make_float2(0.f, 0.f);
    QT[idx] = in ? make_float2(Qre[j * D + i], Qim[j * D + i]) : make_float2(0.f, 0.f);
}

// grad_out: dQ_re [D*D] | dQ_im [D*D] | dR_c_re [D*D] | sum_b loss_b
__global__ void k_finalize_legacy(Dev P, const float* __restrict__ sums, const float* __restrict__ loss,
                                  float* __restrict__ grad_out) {
    const int D = P.D, DP = P.DP, DD = DP * DP;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int idx = tid; idx < D * D; idx += nth) {
        const int o = (idx / D) * DP + idx % D;
        grad_out[idx] = sums[2 * DD + o];
        grad_out[D * D + idx] = sums[3 * DD + o];
        grad_out[2 * D * D + idx] = sums[o];
    }
    if (blockIdx.x == 0 && threadIdx.x < 64) {
        double ls = 0.0;
        for (int b = threadIdx.x; b < P.B; b += 64) ls += (double)loss[b];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) ls += __shfl_xor(ls, off, 64);
        if (threadIdx.x == 0) grad_out[3 * D * D] = (float)ls;
    }
}

// rho = 1 ((N + 1) DP entries), psi_0 = e_0, dtk = 0 (N + 64): the tables the shared wave kernels read
__global__ void k_legacy_tables(int DP, int N, float2* __restrict__ psi0, float* __restrict__ dtk, float2* __restrict__ rho) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = idx; i < (N + 1) * DP; i += nth) rho[i] = make_float2(1.f, 0.f);
    for (int i = idx; i < N + 64; i += nth) dtk[i] = 0.f;
    if (idx < DP) psi0[idx] = make_float2(idx == 0 ? 1.f : 0.f, 0.f);
}

hipError_t launch_legacy_tables(const Dev& P, float2* psi0, float* dtk, float2* rho, hipStream_t s) {
    hipLaunchKernelGGL(k_legacy_tables, dim3(256), dim3(256), 0, s, P.DP, P.N, psi0, dtk, rho);
    return hipGetLastError();
}

hipError_t launch_pack_legacy(const Dev& P, const float* Rr, const float* Qre, const float* Qim, float2* R,
                              float2* RT, float2* Q, float2* QT, hipStream_t s) {
    const int n = P.DP * P.DP;
    hipLaunchKernelGGL(k_pack_legacy, dim3((n + 255) / 256), dim3(256), 0, s, P.D, P.DP, Rr, Qre, Qim, R, RT, Q, QT);
    return hipGetLastError();
}

hipError_t launch_fwd_legacy(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    const size_t shm = (size_t)P.D * sizeof(float2) + 64;
    if (P.D <= 64)
        hipLaunchKernelGGL(k_fwd_legacy<64>, dim3(P.B), dim3(64), shm, s, P, audio, loss, save ? 1 : 0);
    else
        hipLaunchKernelGGL(k_fwd_legacy<128>, dim3(P.B), dim3(128), shm, s, P, audio, loss, save ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_bwd_legacy(const Dev& P, const float* audio, hipStream_t s) {
    const size_t shm = (size_t)3 * P.D * sizeof(float2) + 128;
    if (P.D <= 32)
        hipLaunchKernelGGL((k_bwd_legacy<64, 16>), dim3(P.B), dim3(64), shm, s, P, audio);
    else if (P.D <= 64)
        hipLaunchKernelGGL((k_bwd_legacy<256, 16>), dim3(P.B), dim3(256), shm, s, P, audio);
    else
        hipLaunchKernelGGL((k_bwd_legacy<1024, 16>), dim3(P.B), dim3(1024), shm, s, P, audio);
    return hipGetLastError();
}

hipError_t launch_finalize_legacy(const Dev& P, const float* loss, float* grad_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize_legacy, dim3(8), dim3(256), 0, s, P, (const float*)P.sums, loss, grad_out);
    return hipGetLastError();
}

}  // namespace cmps
