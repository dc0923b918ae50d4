// Legacy `AudioMPS` arithmetic (see cmps_legacy.hip for the recurrence, its adjoint and the
// graph.pbtxt lines), wave-per-clip kernels for D <= 32 on the register-resident mat-vec machinery of the pure-state
// wave kernels (cmps_wave_util.h).  Straight-line float32 code: two wave reductions per forward step (e and |psi'|^2),
// one per reverse step; the two rank-1 gradient terms of a reverse step are three exact float32 MFMAs.
//   forward stash per step and lane: ((R psi) own, psi' own); psi_k itself is psi'_{k-1} / |psi'_{k-1}| (e_0 at k = 0), so
//   the reverse step needs no recomputation and the row is the 512 B of the pure-state wave stash.
#include "cmps_wave_util.h"

namespace cmps {

namespace {

__device__ __forceinline__ void lw_all8(v4f (&o)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]) : : "memory");
}
__device__ __forceinline__ void lbcast(unsigned wr, unsigned rd, float mine, v4f (&q)[8]) {
    bcast_issue(wr, rd, mine, q);
    lw_all8(q);
}

}  // namespace

template <bool SAVE>
__global__ __launch_bounds__(64 * WAVES, 1) void k_fwd_legacy_wave(Dev P, const float* __restrict__ audio,
                                                                   float* __restrict__ loss_out) {
    __shared__ __attribute__((aligned(16))) float2 bcU[WAVES][DPW];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH;
    v2f MR[16], MQ[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        MR[m] = ld2(&P.R[i * DPW + 16 * h + m]);                 // (R[i][j], 0)
        MQ[m] = ld2(&P.Q[i * DPW + 16 * h + m]);
    }
    const unsigned aUw = lds_addr(&bcU[w][0]) + i * 8 + h * 4, aUr = lds_addr(&bcU[w][0]) + h * 128;
    const float* xrow = audio + (size_t)b * T;
    float2* st = SAVE ? reinterpret_cast<float2*>(P.hst) + (size_t)b * N * 64 + lane : nullptr;
    float* sc = SAVE ? P.scal + (size_t)b * NC * 128 : nullptr;
    float psi = (i == 0 && !hb) ? 1.f : 0.f;                     // one_hot(0, D)
    float loss = 0.f;
    v4f q[8];
    for (int c = 0; c < NC; ++c) {
        const int kbeg = c * CH;
        const int cnt = (N - kbeg) < CH ? (N - kbeg) : CH;
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f, x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        const float incv = x1 - x0;
        float nvec = 1.f, evec = 0.f;
        for (int kk = 0; kk < cnt; ++kk) {
            const float x = rdlane(incv, kk);
            const float cdt = P.dt * x;
            lbcast(aUw, aUr, psi, q);
            v2f av, aq;
            mv2_lo(MR, MQ, q, av, aq);
            mv2_hi(MR, MQ, q, av, aq);
            const float v = swapadd(av.x, av.y), qq = swapadd(aq.x, aq.y);
            const float e = 2.0f * sum64(psi * v);               // graph.pbtxt:11857-12661
            const float d = x - e;
            loss += d * d / 2.0f;                                // :12685-12819
            const float y = psi + qq + cdt * v;                  // :12982-14323
            const float n = sum64(y * y);
            nvec = lane == kk ? n : nvec;
            evec = lane == kk ? e : evec;
            if (SAVE) st[(size_t)(kbeg + kk) * 64] = make_float2(v, y);
            psi = y * (1.0f / sqrtf(fmaxf(n, 1e-12f)));          // :14350-14594
        }
        if (SAVE) {
            sc[(size_t)c * 128 + lane] = nvec;
            sc[(size_t)c * 128 + 64 + lane] = evec;
        }
    }
    if (lane == 0) loss_out[b] = loss;
}

__global__ __launch_bounds__(64 * WAVES, 1) void k_bwd_legacy_wave(Dev P, const float* __restrict__ audio) {
    __shared__ __attribute__((aligned(16))) float2 bcB[WAVES][DPW];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH;
    v2f MQd[16], MRt[16];                                        // Q^dagger and R^T rows
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const v2f qt = ld2(&P.QT[i * DPW + 16 * h + m]);          // Q[16h+m][i]
        MQd[m] = mk2(qt.x, -qt.y);
        MRt[m] = ld2(&P.RT[i * DPW + 16 * h + m]);                // (R[16h+m][i], 0)
    }
    const unsigned aBw = lds_addr(&bcB[w][0]) + i * 8 + h * 4, aBr = lds_addr(&bcB[w][0]) + h * 128;
    const float* xrow = audio + (size_t)b * T;
    const float2* st = reinterpret_cast<const float2*>(P.hst) + (size_t)b * N * 64 + lane;
    const float* sc = P.scal + (size_t)b * NC * 128;
    v16f Rre = {}, Qre = {}, Qim = {};
    float g = 0.f;
    v4f q[8];
    for (int c = NC - 1; c >= 0; --c) {
        const int kbeg = c * CH;
        const int cnt = (N - kbeg) < CH ? (N - kbeg) : CH;
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f, x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        const float incv = x1 - x0;
        const float nv = idx < N ? sc[(size_t)c * 128 + lane] : 1.f;
        const float ev = idx < N ? sc[(size_t)c * 128 + 64 + lane] : 0.f;
        const float invv = 1.0f / sqrtf(fmaxf(nv, 1e-12f));
        const float okv = nv > 1e-12f ? 1.f : 0.f;
        const float tev = 2.0f * (ev - incv);
        const float nbelow = kbeg > 0 ? sc[(size_t)(c - 1) * 128 + 63] : 1.f;      // |psi'|^2 of the step below the chunk
        const float invbelow = 1.0f / sqrtf(fmaxf(nbelow, 1e-12f));
        float2 row = st[(size_t)(kbeg + cnt - 1) * 64];
        for (int kk = cnt - 1; kk >= 0; --kk) {
            const int k = kbeg + kk;
            const float2 nxt = st[(size_t)(k > 0 ? k - 1 : 0) * 64];
            const float x = rdlane(incv, kk), inv = rdlane(invv, kk), ok = rdlane(okv, kk), te = rdlane(tev, kk);
            const float invp = kk > 0 ? rdlane(invv, kk - 1) : invbelow;
            const float cdt = P.dt * x;
            const float v = row.x, y = row.y;
            const float p = k > 0 ? nxt.y * invp : ((i == 0 && !hb) ? 1.f : 0.f);     // psi_k
            const float yh = y * inv;
            const float dot = sum64(yh * g);
            const float ybar = (g - ok * yh * dot) * inv;
            const float vbar = cdt * ybar + te * p;
            lbcast(aBw, aBr, ybar, q);
            const v2f aa = mv1(MQd, q);
            const float a = swapadd(aa.x, aa.y);                 // (Q^dagger ybar)
            lbcast(aBw, aBr, vbar, q);
            const v2f ar = mv1(MRt, q);
            const float r = swapadd(ar.x, ar.y);                 // (R^dagger vbar)
            g = ybar + a + te * v + r;
            const float po = osig_of(p, hb);
            Qre = __builtin_amdgcn_mfma_f32_32x32x2f32(ybar, p, Qre, 0, 0, 0);
            Qim = __builtin_amdgcn_mfma_f32_32x32x2f32(ybar, po, Qim, 0, 0, 0);
            Rre = __builtin_amdgcn_mfma_f32_32x32x2f32(vbar, p, Rre, 0, 0, 0);
            row = nxt;
        }
    }
    float* slab = P.slabs + (size_t)b * P.slab_floats;
    constexpr int DD = DPW * DPW;
    for (int idx = lane; idx < (int)P.slab_floats; idx += 64) slab[idx] = 0.f;
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int rw = (rr & 3) + 8 * (rr >> 2) + 4 * h;         // C/D layout of the 32x32 MFMA: column = lane & 31
        const int o = rw * DPW + i;
        slab[o] = Rre[rr];
        slab[2 * DD + o] = Qre[rr];
        slab[3 * DD + o] = -Qim[rr];
    }
}

hipError_t launch_fwd_legacy_wave(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (save)
        hipLaunchKernelGGL(k_fwd_legacy_wave<true>, dim3(nb), dim3(64 * WAVES), 0, s, P, audio, loss);
    else
        hipLaunchKernelGGL(k_fwd_legacy_wave<false>, dim3(nb), dim3(64 * WAVES), 0, s, P, audio, loss);
    return hipGetLastError();
}

hipError_t launch_bwd_legacy_wave(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(k_bwd_legacy_wave, dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    return hipGetLastError();
}

}  // namespace cmps
