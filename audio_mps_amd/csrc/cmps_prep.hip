// Parameter-derived tables, rebuilt once per optimiser step (everything here is batch-independent).
// Reference lines: model.py:16,266,281 (float32 time accumulation), :304-305 (phases), :308 (adjoint),
// :312 (-delta_t sigma^2 / 2), :41-42 (complex R).
#include "cmps_internal.h"

namespace cmps {

// t_0 = 0; t_{k+1} = fl32(t_k + dt), strictly sequential (one lane).  dtk[k] = t_k - t_{k+1} is exact.
__global__ void k_ttable(float dt, int N, float* __restrict__ ttab, float* __restrict__ dtk) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    float t = 0.0f;
    for (int k = 0; k <= N; ++k) {
        ttab[k] = t;
        float tn = __fadd_rn(t, dt);
        if (k < N) dtk[k] = __fsub_rn(t, tn);
        t = tn;
    }
    for (int k = N; k < N + 64; ++k) dtk[k] = 0.0f;  // padding read by the wave kernel's chunked loads
}

// Pack R, R^T, psi0, freqs into DP-strided, zero-padded tables; Q = c_half * R^dagger R (fp64 accumulate,
// rounded once).
__global__ void k_pack(int D, int DP, const float* __restrict__ R_re, const float* __restrict__ R_im,
                       const float* __restrict__ freqs, const float* __restrict__ psi0_re,
                       const float* __restrict__ psi0_im, float c_half, float2* __restrict__ R,
                       float2* __restrict__ RT, float2* __restrict__ Q, float2* __restrict__ psi0,
                       float* __restrict__ freqs_out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < DP * DP) {
        const int i = idx / DP, j = idx % DP;
        float2 r = make_float2(0.f, 0.f), rt = make_float2(0.f, 0.f), q = make_float2(0.f, 0.f);
        if (i < D && j < D) {
            r = make_float2(R_re[i * D + j], R_im[i * D + j]);
            rt = make_float2(R_re[j * D + i], R_im[j * D + i]);
            double qr = 0.0, qi = 0.0;  // (R^dagger R)[i][j] = sum_k conj(R[k][i]) R[k][j]
            for (int k = 0; k < D; ++k) {
                const double ar = R_re[k * D + i], ai = -(double)R_im[k * D + i];
                const double br = R_re[k * D + j], bi = R_im[k * D + j];
                qr += ar * br - ai * bi;
                qi += ar * bi + ai * br;
            }
            q = make_float2((float)((double)c_half * qr), (float)((double)c_half * qi));
        }
        R[idx] = r;
        RT[idx] = rt;
        Q[idx] = q;
    }
    if (idx < DP) {
        const bool in = idx < D;
        psi0[idx] = in ? make_float2(psi0_re[idx], psi0_im[idx]) : make_float2(0.f, 0.f);
        freqs_out[idx] = in ? freqs[idx] : 0.f;
    }
}

// rho_k[d] = phi_k[d] * conj(phi_{k+1}[d]) with phi_k[d] = exp(i * fl32(f_d * t_k))  (model.py:305:
// the argument of the exponential is the float32 product).  The two float32 angles are subtracted
// exactly in double and the rotation is evaluated in double, then rounded once.
__global__ void k_rho(int D, int DP, int N, const float* __restrict__ freqs,
                      const float* __restrict__ ttab, float2* __restrict__ rho) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)(N + 1) * DP) return;
    const int k = (int)(idx / DP), d = (int)(idx % DP);
    float2 out = make_float2(1.f, 0.f);
    if (d < D && k < N) {
        const float f = freqs[d];
        const float th0 = __fmul_rn(f, ttab[k]);
        const float th1 = __fmul_rn(f, ttab[k + 1]);
        const double del = (double)th0 - (double)th1;
        double sn, cs;
        sincos(del, &sn, &cs);
        out = make_float2((float)cs, (float)sn);
    }
    rho[idx] = out;
}

hipError_t launch_prep(const Dev& P, const float* R_re, const float* R_im, const float* freqs,
                       const float* psi0_re, const float* psi0_im, float dt, bool rebuild_ttab,
                       float* ttab, float* dtk, float2* R, float2* RT, float2* Q, float2* psi0,
                       float* freqs_out, float2* rho, hipStream_t s) {
    if (rebuild_ttab) hipLaunchKernelGGL(k_ttable, dim3(1), dim3(64), 0, s, dt, P.N, ttab, dtk);
    const int n = P.DP * P.DP;
    hipLaunchKernelGGL(k_pack, dim3((n + 255) / 256), dim3(256), 0, s, P.D, P.DP, R_re, R_im, freqs,
                       psi0_re, psi0_im, P.c_half, R, RT, Q, psi0, freqs_out);
    const size_t m = (size_t)(P.N + 1) * P.DP;
    hipLaunchKernelGGL(k_rho, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, P.D, P.DP, P.N,
                       freqs_out, ttab, rho);
    return hipGetLastError();
}

}  // namespace cmps
