// Wave-per-clip kernels (the hot path for D <= 32): one 64-lane wavefront owns one clip for the whole
// scan.  The ancilla state and the three D x D matrices (R, R^dagger, Q) live in registers; the only
// HBM traffic is the audio stream (coalesced 256-B chunks), the per-step rotation table (cache
// resident) and, when training, the per-step state stash.
//
// Lane layout (DP = 32, smaller D zero-padded): lane l = (i = l & 31, h = l >> 5).
//   * every lane holds component i of each state vector as one float2 (both halves hold the same value)
//   * lane (i, h) holds columns 16h .. 16h+15 of row i of R, R^dagger and Q  (3 x 16 float2)
//   * a mat-vec is: broadcast the vector through LDS (each half reads its 16 entries with ds_read_b128),
//     16 complex FMAs per lane, one cross-half add.
// Recurrence and adjoint: see the header of cmps_block.hip (same arithmetic, same reference lines).
#include "cmps_internal.h"

namespace cmps {

namespace {

constexpr int DPW = 32;        // padded bond dimension of this variant
constexpr int WAVES = 4;       // waves (clips) per workgroup: one per SIMD of a CU

__device__ __forceinline__ float rdlane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// sum over the 32 lanes of each half (both halves hold the same values -> same result everywhere)
__device__ __forceinline__ float sum32(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    return v;
}

__device__ __forceinline__ float2 xhalf_add(float2 v) {  // add the partner half's partial sum
    return make_float2(v.x + __shfl_xor(v.x, 32, 64), v.y + __shfl_xor(v.y, 32, 64));
}

// write one vector (component i from the lanes of half 0) and read back this half's 16 entries
__device__ __forceinline__ void bcast16(float2* buf, float2 mine, int i, int h, float2 (&out)[16]) {
    if (h == 0) buf[i] = mine;
    __builtin_amdgcn_wave_barrier();
    const float4* p = reinterpret_cast<const float4*>(buf + 16 * h);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float4 t = p[q];
        out[2 * q] = make_float2(t.x, t.y);
        out[2 * q + 1] = make_float2(t.z, t.w);
    }
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float2 mv16(const float2 (&M)[16], const float2 (&v)[16]) {
    float2 acc0 = make_float2(0.f, 0.f), acc1 = make_float2(0.f, 0.f);
#pragma unroll
    for (int m = 0; m < 16; m += 2) {
        acc0 = cfma(M[m], v[m], acc0);
        acc1 = cfma(M[m + 1], v[m + 1], acc1);
    }
    return cadd(acc0, acc1);
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES, 1) void k_fwd_wave(Dev P, const float* __restrict__ audio,
                                                            float* __restrict__ loss_out, int save) {
    __shared__ __attribute__((aligned(16))) float2 lds[WAVES][2][DPW];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;  // whole wave exits together; no workgroup barriers are used below
    const int N = P.N, T = P.T;
    float2* bufU = lds[w][0];
    float2* bufY = lds[w][1];

    float2 MR[16], MQ[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        MR[m] = P.R[i * DPW + 16 * h + m];
        MQ[m] = P.Q[i * DPW + 16 * h + m];
    }
    const float* xrow = audio + (size_t)b * T;
    float2* st = save ? P.stash + (size_t)b * N * DPW : nullptr;
    float2 u = P.psi0[i];
    float loss = 0.f;
    float incv = 0.f, sv = 0.f;
    float2 rho_next = P.rho[i];
    for (int k = 0; k < N; ++k) {
        if ((k & 63) == 0) {  // next 64 increments, one per lane (model.py:263, :303)
            const int idx = k + lane;
            const float a0 = idx < T ? xrow[idx] : 0.f;
            const float a1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            incv = a1 - a0;
            sv = incv / P.A;
        }
        const float x = rdlane(incv, k & 63);
        const float s = rdlane(sv, k & 63);
        const float2 rho = rho_next;
        rho_next = P.rho[(size_t)(k + 1) * DPW + i];  // table has N+1 rows
        float2 ub[16];
        bcast16(bufU, u, i, h, ub);
        const float2 v = xhalf_add(mv16(MR, ub));
        const float2 q = xhalf_add(mv16(MQ, ub));
        const float2 y = make_float2(u.x + q.x + s * v.x, u.y + q.y + s * v.y);
        float2 yb[16];
        bcast16(bufY, y, i, h, yb);
        const float2 r = xhalf_add(mv16(MR, yb));
        const float e = 2.0f * sum32(y.x * r.x + y.y * r.y);
        const float n = sum32(y.x * y.x + y.y * y.y);
        const float z = (e * x) / P.A;
        loss += -logf(1.0f + z);
        if (st && h == 0) st[(size_t)k * DPW + i] = y;
        const float inv = 1.0f / sqrtf(fmaxf(n, 1e-12f));
        u = cmul(rho, cscale(inv, y));
    }
    if (lane == 0) loss_out[b] = loss;
}

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WAVES, 1) void k_bwd_wave(Dev P, const float* __restrict__ audio) {
    __shared__ __attribute__((aligned(16))) float2 lds[WAVES][3][DPW];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;
    const int N = P.N, T = P.T;
    float2* bufY = lds[w][0];
    float2* bufYb = lds[w][1];
    float2* bufU = lds[w][2];
    const float2 zero = make_float2(0.f, 0.f);

    float2 MR[16], MRd[16], MQ[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        MR[m] = P.R[i * DPW + 16 * h + m];
        const float2 rt = P.RT[i * DPW + 16 * h + m];  // R[16h+m][i]
        MRd[m] = make_float2(rt.x, -rt.y);
        MQ[m] = P.Q[i * DPW + 16 * h + m];
    }
    float2 Rb[16], Qb[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) Rb[m] = Qb[m] = zero;

    const float* xrow = audio + (size_t)b * T;
    const float2* st = P.stash + (size_t)b * N * DPW;
    const float A = P.A, invA2 = 1.0f / (A * A);
    float facc = 0.f, Abar = 0.f;
    float2 g = zero;

    float2 y = st[(size_t)(N - 1) * DPW + i];
    float nraw = sum32(y.x * y.x + y.y * y.y);
    float inv = 1.0f / sqrtf(fmaxf(nraw, 1e-12f));
    float2 yhat = cscale(inv, y);
    float2 rho = P.rho[(size_t)(N - 1) * DPW + i];
    float2 unext = cmul(rho, yhat);
    // prefetched for the step's second half: y_{k-1}, rho_{k-1}
    float2 yprev_n = N >= 2 ? st[(size_t)(N - 2) * DPW + i] : zero;
    float2 rhoprev_n = N >= 2 ? P.rho[(size_t)(N - 2) * DPW + i] : make_float2(1.f, 0.f);

    float incv = 0.f, sv = 0.f, dtv = 0.f;
    int chunk = -1;
    for (int k = N - 1; k >= 0; --k) {
        if ((k >> 6) != chunk) {
            chunk = k >> 6;
            const int idx = (chunk << 6) + lane;
            const float a0 = idx < T ? xrow[idx] : 0.f;
            const float a1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            incv = a1 - a0;
            sv = incv / A;
            dtv = P.dtk[idx];  // padded to N + 64 entries
        }
        const float x = rdlane(incv, k & 63);
        const float s = rdlane(sv, k & 63);
        const float dtk = rdlane(dtv, k & 63);
        const float2 yprev = yprev_n, rhoprev = rhoprev_n;
        if (k >= 2) {
            yprev_n = st[(size_t)(k - 2) * DPW + i];
            rhoprev_n = P.rho[(size_t)(k - 2) * DPW + i];
        }
        facc += dtk * (g.y * unext.x - g.x * unext.y);
        const float2 yhb = cmul_conj_a(rho, g);
        const float dot = sum32(yhat.x * yhb.x + yhat.y * yhb.y);
        float2 ybar;
        if (nraw > 1e-12f)
            ybar = make_float2((yhb.x - yhat.x * dot) * inv, (yhb.y - yhat.y * dot) * inv);
        else
            ybar = cscale(inv, yhb);
        float2 yb[16];
        bcast16(bufY, y, i, h, yb);
        const float2 r = xhalf_add(mv16(MR, yb));
        const float2 a = xhalf_add(mv16(MRd, yb));
        const float e = 2.0f * sum32(y.x * r.x + y.y * r.y);
        const float ex = e * x;
        const float z = ex / A;
        const float zbar = -1.0f / (1.0f + z);
        const float ebar = zbar * x / A;
        Abar += zbar * (-ex * invA2);
        const float te = 2.0f * ebar;
        ybar.x += te * (r.x + a.x);
        ybar.y += te * (r.y + a.y);
        float2 ybb[16];
        bcast16(bufYb, ybar, i, h, ybb);
        const float2 bq = xhalf_add(mv16(MQ, ybb));
        const float2 d = xhalf_add(mv16(MRd, ybb));
        // u_k
        float2 uk, yhatp = zero;
        float nprev = 1.f, invp = 1.f;
        if (k > 0) {
            nprev = sum32(yprev.x * yprev.x + yprev.y * yprev.y);
            invp = 1.0f / sqrtf(fmaxf(nprev, 1e-12f));
            yhatp = cscale(invp, yprev);
            uk = cmul(rhoprev, yhatp);
        } else {
            uk = P.psi0[i];
        }
        const float sbar = sum32(d.x * uk.x + d.y * uk.y);
        Abar += sbar * (-x * invA2);
        float2 ukb[16];
        bcast16(bufU, uk, i, h, ukb);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 yj = yb[m], uj = ukb[m];
            const float2 o1 = make_float2(y.x * yj.x + y.y * yj.y, y.y * yj.x - y.x * yj.y);
            const float2 o2 = make_float2(ybar.x * uj.x + ybar.y * uj.y, ybar.y * uj.x - ybar.x * uj.y);
            Rb[m].x += te * o1.x + s * o2.x;
            Rb[m].y += te * o1.y + s * o2.y;
            Qb[m].x += o2.x;
            Qb[m].y += o2.y;
        }
        g = make_float2(ybar.x + bq.x + s * d.x, ybar.y + bq.y + s * d.y);
        y = yprev; nraw = nprev; inv = invp; yhat = yhatp; unext = uk; rho = rhoprev;
    }
    float* slab = P.slabs + (size_t)b * P.slab_floats;
    constexpr int DD = DPW * DPW;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int o = i * DPW + 16 * h + m;
        slab[o] = Rb[m].x;
        slab[DD + o] = Rb[m].y;
        slab[2 * DD + o] = Qb[m].x;
        slab[3 * DD + o] = Qb[m].y;
    }
    if (h == 0) {
        slab[4 * DD + i] = facc;
        slab[4 * DD + DPW + i] = g.x;
        slab[4 * DD + 2 * DPW + i] = g.y;
    }
    if (lane == 0) {
        slab[4 * DD + 3 * DPW] = Abar;
        slab[4 * DD + 3 * DPW + 1] = 0.f;
    }
}

hipError_t launch_fwd_wave(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(k_fwd_wave, dim3(nb), dim3(64 * WAVES), 0, s, P, audio, loss, save ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_bwd_wave(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(k_bwd_wave, dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    return hipGetLastError();
}

}  // namespace cmps
