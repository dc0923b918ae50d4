// Two wavefronts per clip (D <= 32): the forward scan split into a CHAIN wave and a LOSS wave.
//
// Why: a lone wave issues one VALU instruction per ~5.4 cycles, two waves on one SIMD issue one per ~4.1 cycles
// combined (profiles/r1_ubench_two_waves.log), and at B = 1024 clips there is exactly one clip per SIMD.  The forward
// step has a serial part (u -> y = u + Q u + s R u -> |y|^2 -> u') and a part nothing waits for (H y, e = y^dagger H y,
// the stash row, the loss).  k_fwd_wave runs the second one step late inside the same wave; here it runs in a second
// wave that shares the SIMDs with the chain waves and fills their idle issue slots:
//
//   chain wave (waves 0-3 of the workgroup, raised priority): per step two interleaved mat-vec chains, one wave
//     reduction, the rotation; writes y_k (256 B, split layout) into a ring in LDS; per 32-step chunk publishes
//     `prod` = chunks written and stores the chunk's |y_k|^2 row.
//   loss wave (waves 4-7): per step reads y_k back (own value + this half's 16 entries), H y_k, the per-lane product for
//     e_k, the stash row (y_k, H y_k); per 32-step chunk the column sums, log(1 + e x / A) in the reference's operation
//     order and the sequential float32 loss accumulation (model.py:279, 294); publishes `cons` = chunks consumed.
//
// One mat-vec on the chain, not two: y = ut + Q ut + s_k R ut = ut + M_k ut with M_k = Q + s_k R.  Forming M_k costs one
// packed FMA per complex entry (16 per lane) against the two a second mat-vec costs, and it does not depend on the state:
// M_{k+1} is formed in the shadow of step k's LDS broadcast, where the in-order wave would otherwise wait.
//
// Synchronisation is two LDS counters per clip and NO barrier: the ring holds two chunks; the loss wave starts chunk
// c when prod >= c + 1, the chain wave starts chunk c (c >= 2) when cons >= c - 1.  LDS operations of one wave
// complete in order, so a counter write issued after the chunk's data is visible after it.
// Same arithmetic as k_fwd_wave (cmps_wave.hip), same stash layouts; the reverse sweep is unchanged.
#include "cmps_wave_util.h"

namespace cmps {

namespace {

constexpr int CH2 = 32;          // steps per chunk (ring half, rho staging, per-chunk scalar math)
constexpr int RING = 2 * CH2;    // ring slots
constexpr int PE2_LD = 33;       // row stride of the loss wave's product buffer

// progress counters in LDS, accessed with explicit DS instructions (a `volatile int*` cast would decay to a generic
// pointer: flat accesses plus a vmcnt(0) wait that also drains the stash stores)
__device__ __forceinline__ int flag_load(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void flag_store(unsigned addr, int v, int lane) {
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

// this half's 16 entries of a ring row and the lane's own value
__device__ __forceinline__ void rows_own_issue(unsigned rd, unsigned own, v4f (&o)[8], float& mine) {
    asm volatile("ds_read_b128 %0, %9\n\tds_read_b128 %1, %9 offset:16\n\t"
                 "ds_read_b128 %2, %9 offset:32\n\tds_read_b128 %3, %9 offset:48\n\t"
                 "ds_read_b128 %4, %9 offset:64\n\tds_read_b128 %5, %9 offset:80\n\t"
                 "ds_read_b128 %6, %9 offset:96\n\tds_read_b128 %7, %9 offset:112\n\t"
                 "ds_read_b32 %8, %10"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]),
                   "=&v"(mine)
                 : "v"(rd), "v"(own) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait_own9(v4f (&o)[8], float& mine) {
    asm volatile("s_waitcnt lgkmcnt(%9)"
                 : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(mine)
                 : "n"(N) : "memory");
}

// One mat-vec chain (this half's 16 columns of M u) with the wave reduction of `x` threaded through it: a lone in-order
// wave pays ~20 cycles per dependent DPP step, but nothing when four independent packed FMAs sit between two steps.
// After mv1r_hi, `tot` (SGPR) holds the sum of x over the 64 lanes.
#define DPPADD(ctrl) "v_add_f32_dpp %1, %1, %1 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void mv1r_lo(const v2f (&M)[16], const v4f (&q)[8], v2f& acc, float& x) {
    asm(CM_FIRST(0, 2, 10) CM(0, 3, 11) DPPADD("quad_perm:[1,0,3,2]")
        CM(0, 4, 12) CM(0, 5, 13) DPPADD("quad_perm:[2,3,0,1]")
        CM(0, 6, 14) CM(0, 7, 15) DPPADD("row_half_mirror")
        CM(0, 8, 16) CM(0, 9, 17) DPPADD("row_mirror")
        : "=&v"(acc), "+v"(x)
        : "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(M[4]), "v"(M[5]), "v"(M[6]), "v"(M[7]),
          "v"(lo2(q[0])), "v"(hi2(q[0])), "v"(lo2(q[1])), "v"(hi2(q[1])), "v"(lo2(q[2])), "v"(hi2(q[2])),
          "v"(lo2(q[3])), "v"(hi2(q[3])));
}
__device__ __forceinline__ void mv1r_hi(const v2f (&M)[16], const v4f (&q)[8], v2f& acc, float& x, float& tot) {
    asm(CM(0, 3, 11) CM(0, 4, 12) CM(0, 5, 13)
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        CM(0, 6, 14) CM(0, 7, 15)
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        CM(0, 8, 16) CM(0, 9, 17)
        "v_readlane_b32 %2, %1, 63\n\t"
        CM(0, 10, 18)
        : "+v"(acc), "+v"(x), "=s"(tot)
        : "v"(M[8]), "v"(M[9]), "v"(M[10]), "v"(M[11]), "v"(M[12]), "v"(M[13]), "v"(M[14]), "v"(M[15]),
          "v"(lo2(q[4])), "v"(hi2(q[4])), "v"(lo2(q[5])), "v"(hi2(q[5])), "v"(lo2(q[6])), "v"(hi2(q[6])),
          "v"(lo2(q[7])), "v"(hi2(q[7])));
}
#undef DPPADD
// wait for the second half of a broadcast; `dep` ties the wait behind the first FMA block (a plain asm statement the
// scheduler would otherwise be free to sink below this volatile one)
template <int N>
__device__ __forceinline__ void lds_wait_hi_t_after(v4f (&o)[8], v2f& t, v2f& dep) {
    asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]), "+v"(t), "+v"(dep) : "n"(N) : "memory");
}
// vec[lane sel] = val (both wave-uniform): v_writelane_b32 takes its lane select from M0 when the value already occupies the
// one SGPR operand slot; M0 is reserved by the compiler, so it is saved and restored around the instruction
__device__ __forceinline__ void write_lane(float& vec, float val, int sel) {
    unsigned keep;
    asm("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
        : "+v"(vec), "=&s"(keep) : "s"(val), "s"(sel));
}
__device__ __forceinline__ float vmax_s(float s, float c) {     // one v_max_f32 (fmaxf adds a canonicalising second one)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "s"(s), "v"(c));
    return r;
}

}  // namespace

// Diagnostic builds only (scripts/ablate.py passes -DCMPS_DIAG -DCMPS_DIAG_NO_LOSS / -DCMPS_DIAG_NO_CHAIN; results are wrong,
// the timing tells what each wave costs alone).  tests/test_capi_load.py compiles both so that they cannot rot.
#if defined(CMPS_DIAG) && defined(CMPS_DIAG_NO_LOSS)
#define DIAG_NO_LOSS 1
#else
#define DIAG_NO_LOSS 0
#endif
#if defined(CMPS_DIAG) && defined(CMPS_DIAG_NO_CHAIN)
#define DIAG_NO_CHAIN 1
#else
#define DIAG_NO_CHAIN 0
#endif

template <bool SAVE>
__global__ __launch_bounds__(128 * WAVES, 1) void k_fwd_wave2(Dev P, const float* __restrict__ audio,
                                                              float* __restrict__ loss_out) {
    __shared__ __attribute__((aligned(16))) float4 stR[WAVES][CH2 * 16];   // rho rows of the chain wave's chunk
    __shared__ __attribute__((aligned(16))) float2 bcU[WAVES][DPW];
    __shared__ __attribute__((aligned(16))) float2 ring[WAVES][RING][DPW];  // y_k, interleaved (re, im) per component
    __shared__ float pe[WAVES][64 * PE2_LD];
    __shared__ int flags[WAVES][2];                                         // [clip][0: prod, 1: cons]
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int w = wv & (WAVES - 1), role = wv / WAVES;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    if (threadIdx.x < 2 * WAVES) (&flags[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x * WAVES + w;
    if (b >= P.B) return;      // both waves of the clip leave together
    const int N = P.N, T = P.T, NC2 = (N + CH2 - 1) / CH2, NC = (N + CH - 1) / CH;
    const float* xrow = audio + (size_t)b * T;
    const float A = P.A;
    float* sc = SAVE ? P.scal + scal_off(b, NC, 0) : nullptr;
    const unsigned aRing = lds_addr(&ring[w][0][0]);
    const unsigned aProd = lds_addr(&flags[w][0]), aCons = lds_addr(&flags[w][1]);

    if (role == 0) {
        // ------------------------------------------------------------------ chain wave
        if (DIAG_NO_CHAIN) return;
        __builtin_amdgcn_s_setprio(3);
        stagger(w);
        v2f MR[16], MQ[16], MM[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            MR[m] = ld2(&P.R[i * DPW + 16 * h + m]);
            MQ[m] = ld2(&P.Q[i * DPW + 16 * h + m]);
        }
        const unsigned aUw = lds_addr(&bcU[w][0]) + i * 8 + h * 4, aUr = lds_addr(&bcU[w][0]) + h * 128;
        const unsigned aYw = aRing + i * 8 + h * 4;
        const unsigned aRho = lds_addr(&stR[w][0]) + i * 8;
        const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
        v4f sr[8];
        stage_load<8>(rho4, 0, N, lane, sr);
        float xa0 = lane < T ? xrow[lane] : 0.f;
        float xa1 = lane + 1 < T ? xrow[lane + 1] : 0.f;
        stage_commit<8>(stR[w], lane, sr);
        const float2 p0 = P.psi0[i];
        float u = hb ? p0.y : p0.x;
        v4f qu[8];
        v2f rho;
        // The normalisation is linear, so it is applied AFTER the mat-vec: the wave broadcasts ut = rho_{k-1} y_{k-1}
        // (un-normalised), y_k = inv_{k-1} (ut + M_k ut) with inv_{k-1} = rsqrt(max(|y_{k-1}|^2, 1e-12)), and the
        // reduction of |y_{k-1}|^2 rides inside the FMA blocks of step k instead of sitting on the serial chain.
        float xsq = lane == 0 ? 1.f : 0.f;      // "|y_{-1}|^2" = 1: psi_0 arrives normalised
        float nvec = 1.f;
        float sv = (xa1 - xa0) / A;             // model.py:263, 303: s_k = x_k / A, one step per lane
        // M_k = Q + s_k R (model.py:308-313 with the two products merged): one packed FMA per complex entry
#define FORM_M(S_)                                                                                            \
        {                                                                                                     \
            const float s_ = (S_);                                                                            \
            const v2f s2_ = mk2(s_, s_);                                                                      \
            _Pragma("unroll") for (int m = 0; m < 16; ++m) MM[m] = __builtin_elementwise_fma(MR[m], s2_, MQ[m]); \
        }
        FORM_M(rdlane(sv, 0))
        bcast_issue_tab(aUw, aUr, u, aRho, qu, rho);
        for (int c = 0; c < NC2; ++c) {
            const int kbeg = c * CH2;
            const int cnt = (N - kbeg) < CH2 ? (N - kbeg) : CH2;
            {
                const int cn = c + 1 < NC2 ? c + 1 : NC2 - 1;
                stage_load<8>(rho4, cn * CH2, N, lane, sr);
                const int idx = cn * CH2 + lane;
                xa0 = idx < T ? xrow[idx] : 0.f;
                xa1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            }
            if (!DIAG_NO_LOSS && c >= 2)                              // the ring half about to be overwritten
                while (flag_load(aCons) < c - 1) __builtin_amdgcn_s_sleep(1);
            unsigned ay = aYw + (c & 1) * (CH2 * 256);
#define CHAIN_STEP(KK)                                                                                        \
            {                                                                                                 \
                const int kk_ = (KK);                                                                         \
                lds_wait_lo<5>(qu);                                                                           \
                v2f am;                                                                                       \
                float nprev;                                                                                  \
                mv1r_lo(MM, qu, am, xsq);                                                                     \
                lds_wait_hi_t_after<0>(qu, rho, am);                                                          \
                mv1r_hi(MM, qu, am, xsq, nprev);                          /* nprev = |y_{k-1}|^2 */           \
                const float inv = __builtin_amdgcn_rsqf(vmax_s(nprev, 1e-12f));  /* model.py:332 */           \
                const float y = inv * (u + swapadd(am.x, am.y));                                              \
                lds_write32(ay, y);                                                                           \
                ay += 256;                                                                                    \
                const float yo = osig_of(y, hb);                                                              \
                const v2f un = cmul2(mk2(y, yo), rho);                    /* rho_k y_k, normalised next step */ \
                u = un.x;                                                                                     \
                const int kn = kk_ + 1 < CH2 ? kk_ + 1 : 0;               /* chunk end: a dummy, retired below */ \
                bcast_issue_tab(aUw, aUr, u, aRho + kn * 256, qu, rho);                                       \
                FORM_M(rdlane(sv, kn))                                    /* in the shadow of the broadcast */ \
                xsq = y * y;                                                                                  \
                write_lane(nvec, nprev, (kk_ - 1) & (CH2 - 1));                                               \
            }
            CHAIN_STEP(0)
            if (SAVE && c > 0 && lane < CH2) sc[(size_t)((c - 1) >> 1) * 128 + ((c - 1) & 1) * CH2 + lane] = nvec;
            for (int kk = 1; kk < cnt; ++kk) CHAIN_STEP(kk)
#undef CHAIN_STEP
            lds_wait_hi_t<0>(qu, rho);                                 // everything of this chunk has landed
            flag_store(aProd, c + 1, lane);                            // publish (ordered behind the chunk's y rows)
            if (c + 1 < NC2) {
                stage_commit<8>(stR[w], lane, sr);
                sv = (xa1 - xa0) / A;                                  // the next chunk's s_k
                FORM_M(rdlane(sv, 0))                                  // the last in-loop FORM_M used the stale lane 0
                bcast_issue_tab(aUw, aUr, u, aRho, qu, rho);
            }
        }
#undef FORM_M
        if (SAVE) {                                                    // |y_{N-1}|^2 closes the last row
            const float nlast = sum64(xsq);
            const int cl = NC2 - 1;
            write_lane(nvec, nlast, (N - 1) & (CH2 - 1));
            if (lane < CH2) sc[(size_t)(cl >> 1) * 128 + (cl & 1) * CH2 + lane] = nvec;
        }
        return;
    }

    // ---------------------------------------------------------------------- loss wave
    if (DIAG_NO_LOSS) return;
    __builtin_amdgcn_s_setprio(0);
    v2f MH[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const v2f r = ld2(&P.R[i * DPW + 16 * h + m]);
        const v2f rt = ld2(&P.RT[i * DPW + 16 * h + m]);   // R[16h+m][i]
        MH[m] = mk2(r.x + rt.x, r.y - rt.y);                // (R + R^dagger)[i][16h+m]
    }
    const unsigned aYr = aRing + h * 128, aYo = aRing + i * 8 + h * 4;
    const unsigned aPEw = lds_addr(&pe[w][0]) + lane * (PE2_LD * 4);
    float2* st = SAVE ? reinterpret_cast<float2*>(P.hst + (size_t)b * N * 128) + lane : nullptr;
    float loss = 0.f;
    v4f qa[8], qb[8];
    float ya = 0.f, yb = 0.f;
    for (int c = 0; c < NC2; ++c) {
        const int kbeg = c * CH2;
        const int cnt = (N - kbeg) < CH2 ? (N - kbeg) : CH2;
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f;
        const float x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        if (!DIAG_NO_CHAIN)
            while (flag_load(aProd) < c + 1) __builtin_amdgcn_s_sleep(1);
        const unsigned off = (c & 1) * (CH2 * 256);
        // Branch-free inner loop (branches around the counted waits make hipcc copy the 64 staging registers): every
        // step issues the reads of the next one, clamped to the chunk's last row; an odd chunk ends with one repeated
        // (idempotent) step, and the trailing dummy reads are retired after the loop.
        const int last = cnt - 1;
        rows_own_issue(aYr + off, aYo + off, qa, ya);
#define LOSS_STEP(KK, Q, Y, QN, YN)                                                                     \
        {                                                                                              \
            const int kk_ = (KK) < last ? (KK) : last;                                                 \
            const int kn_ = (KK) + 1 < last ? (KK) + 1 : last;                                         \
            rows_own_issue(aYr + off + kn_ * 256, aYo + off + kn_ * 256, QN, YN);                      \
            lds_wait_own9<9>(Q, Y);                                                                    \
            const v2f ah = mv1(MH, Q);                                                                 \
            const float hs = swapadd(ah.x, ah.y);                                                      \
            lds_write32(aPEw + kk_ * 4, Y * hs);                                                       \
            if (SAVE) st[(size_t)(kbeg + kk_) * 64] = make_float2(Y, hs);                              \
        }
        for (int kk = 0; kk < cnt; kk += 2) {
            LOSS_STEP(kk, qa, ya, qb, yb)
            LOSS_STEP(kk + 1, qb, yb, qa, ya)
        }
#undef LOSS_STEP
        lds_wait_own9<0>(qa, ya);
        flag_store(aCons, c + 1, lane);                                // every ring read has landed: the half is free
        // e_k = sum over the 64 lanes of the stored products: lane (i, h) sums rows 32h..32h+31 of column i
        float evec;
        {
            const float* col = &pe[w][(32 * h) * PE2_LD + i];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int l = 0; l < 32; l += 4) {
                a0 += col[(l + 0) * PE2_LD];
                a1 += col[(l + 1) * PE2_LD];
                a2 += col[(l + 2) * PE2_LD];
                a3 += col[(l + 3) * PE2_LD];
            }
            const float part = (a0 + a1) + (a2 + a3);
            evec = swapadd(part, part);
        }
        const float incv = x1 - x0;
        const float z = (evec * incv) / A;                             // model.py:294 operation order
        const float lv = -logf(1.0f + z);
        for (int j = 0; j < cnt; ++j) loss += rdlane(lv, j);           // model.py:279: sequential in time
        if (SAVE && lane < CH2) sc[(size_t)(c >> 1) * 128 + 64 + (c & 1) * CH2 + lane] = evec;
    }
    if (lane == 0) loss_out[b] = loss;
}

hipError_t launch_fwd_wave2(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (save)
        hipLaunchKernelGGL(k_fwd_wave2<true>, dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    else
        hipLaunchKernelGGL(k_fwd_wave2<false>, dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    return hipGetLastError();
}

}  // namespace cmps
