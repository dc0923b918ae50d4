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
//   loss wave (waves 4-7): nothing waits for it, so it works a whole 32-step chunk at a time ON THE MATRIX CORES: the 32 rows
//     y_k of the chunk form Y [32 steps x 64 reals], and H Y^T is one real GEMM Y W (W = the 64 x 64 real form of
//     H = R + R^dagger) = 2 tiles x 4 k-steps of v_mfma_f32_32x32x16_bf16.  Both operands are split EXACTLY into three bf16
//     pieces (8 + 8 + 8 significand bits, by truncation) and six products per pair are accumulated in fp32
//     (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid: 24 operand bits, what is dropped is <= 2^-23 |a||b|), so H y_k is
//     fp32-faithful; 48 MFMAs per chunk replace 32 x 32 packed FMAs plus 9 LDS reads per step.  Then e_k = y_k . (H y_k)
//     (products transposed through LDS, column sums), the stash rows (y_k, H y_k), log(1 + e x / A) in the reference's
//     operation order and the sequential float32 loss accumulation (model.py:279, 294); publishes `cons`.
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
constexpr int RLD = 68;          // floats per ring row: 64 (y_k as n = 2 i + {re, im}) + 4 of padding, so that the 32 rows the
                                 // loss wave reads at one column offset (ds_read_b128, 16-lane groups) fall into distinct banks
constexpr int PE2_LD = 36;       // row stride (floats) of the loss wave's product buffer [step][column]: 16-B aligned rows, the
                                 // 32 rows read at one offset fall into distinct banks

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
template <int SEL>
__device__ __forceinline__ void write_lane_c(float& vec, float val) {      // vec[lane SEL] = val, SEL a compile-time constant
    asm("v_writelane_b32 %0, %1, %2" : "+v"(vec) : "s"(val), "n"(SEL));
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

// LEGACY: the arithmetic of the previous-generation AudioMPS (cmps_legacy.hip: the recurrence and its graph.pbtxt lines) on the same
// machinery.  With rho = 1 and psi_0 = e_0 in the tables (cmps_legacy_set_params) the chain is the same linear step,
// y_k = inv_{k-1} (y_{k-1} + M_k y_{k-1}), M_k = Q + dt x_k R; what differs is the loss wave: e_k = psi_k^dagger H psi_k is taken on the
// normalised state BEFORE the update, i.e. e_k = (y_{k-1}^dagger H y_{k-1}) / max(|y_{k-1}|^2, 1e-12) (e_0 = H_00), and
// loss += (x_k - e_k)^2 / 2.  The stash rows (y_k, H y_k) and the |y_k|^2 rows are the same; the e rows hold the legacy e_k.
// HF16 (round 5; what CMPS_RANK1_F16X2 / DEFAULT select for the PsiCMPS arithmetic): the loss wave's GEMM with two power-of-two scaled fp16
// pieces per operand and three products (hi hi + hi lo + lo hi on v_mfma_f32_32x32x16_f16) instead of three bf16 pieces and six: the
// arithmetic of the wide family's k_hy_wide<f16x2>.  Scales: W from its largest entry (below 2^15), the chunk's y rows from
// the guaranteed bound |y_k|_2 <= 1 + |Q|_F + max_chunk |s_k| |R|_F (below 2^14); one exact multiply takes them out of the accumulators.
template <bool SAVE, bool LEGACY = false, bool HF16 = false>
__global__ __launch_bounds__(128 * WAVES, 1) void k_fwd_wave2(Dev P, const float* __restrict__ audio,
                                                              float* __restrict__ loss_out) {
    // rho rows of the chain wave's chunk, two buffers (round 5): the next chunk's rows are fetched in two halves of 16 rows and committed
    // into the OTHER buffer while the current one is being read, so only 16 registers ride through the chunk instead of 32 -- with 32
    // hipcc spilled four of them to scratch memory and reloaded them behind an s_waitcnt vmcnt(0) at every chunk end (ISA of round 4)
    __shared__ __attribute__((aligned(16))) float4 stR[WAVES][2][CH2 * 16];
    __shared__ __attribute__((aligned(16))) float2 bcU[WAVES][DPW];
    __shared__ __attribute__((aligned(16))) float ring[WAVES][RING][RLD];   // y_k, n = 2 i + {re, im}, rows padded (RLD)
    __shared__ __attribute__((aligned(16))) float pe[WAVES][CH2 * PE2_LD]; // y_k[n] (H y_k)[n], [step][column]
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
    const float A = dev_A(P);
    float* sc = SAVE ? P.scal + scal_off(b, NC, 0) : nullptr;
    const unsigned aRing = lds_addr(&ring[w][0][0]);
    constexpr int RROWB = RLD * 4;                                          // bytes per ring row
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
        unsigned aRho = lds_addr(&stR[w][0][0]) + i * 8;
        const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
        v4f sr[4];
        {
            v4f s8[8];
            stage_load<8>(rho4, 0, N, lane, s8);
            stage_commit<8>(stR[w][0], lane, s8);
        }
        float xa0 = lane < T ? xrow[lane] : 0.f;
        float xa1 = lane + 1 < T ? xrow[lane + 1] : 0.f;
        const float2 p0 = P.psi0[i];
        float u = hb ? p0.y : p0.x;
        v4f qu[8];
        v2f rho;
        // The normalisation is linear, so it is applied AFTER the mat-vec: the wave broadcasts ut = rho_{k-1} y_{k-1}
        // (un-normalised), y_k = inv_{k-1} (ut + M_k ut) with inv_{k-1} = rsqrt(max(|y_{k-1}|^2, 1e-12)), and the
        // reduction of |y_{k-1}|^2 rides inside the FMA blocks of step k instead of sitting on the serial chain.
        float xsq = lane == 0 ? 1.f : 0.f;      // "|y_{-1}|^2" = 1: psi_0 arrives normalised
        float nvec = 1.f;
        float sv = LEGACY ? P.dt * (xa1 - xa0) : (xa1 - xa0) / A;   // model.py:263, 303: s_k = x_k / A, one step per lane (legacy: dt x_k)
        // M_k = Q + s_k R (model.py:308-313 with the two products merged): one packed FMA per complex entry
#define FORM_M(S_)                                                                                            \
        {                                                                                                     \
            const float s_ = (S_);                                                                            \
            const v2f s2_ = mk2(s_, s_);                                                                      \
            _Pragma("unroll") for (int m = 0; m < 16; ++m) MM[m] = __builtin_elementwise_fma(MR[m], s2_, MQ[m]); \
        }
        FORM_M(rdlane(sv, 0))
        bcast_issue_tab_sync(aUw, aUr, u, aRho, qu, rho);
        for (int c = 0; c < NC2; ++c) {
            const int kbeg = c * CH2;
            const int cnt = (N - kbeg) < CH2 ? (N - kbeg) : CH2;
            const int cn = c + 1 < NC2 ? c + 1 : NC2 - 1;
            float4* nxt = stR[w][(c + 1) & 1];                         // the buffer the next chunk will read
            {
                stage_load<4>(rho4, cn * CH2, N, lane, sr);            // rows 0 .. 15 of the next chunk
                const int idx = cn * CH2 + lane;
                xa0 = idx < T ? xrow[idx] : 0.f;
                xa1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            }
            if (!DIAG_NO_LOSS && c >= 2)                              // the ring half about to be overwritten
                while (flag_load(aCons) < c - 1) __builtin_amdgcn_s_sleep(1);
            unsigned ay = aYw + (c & 1) * (CH2 * RROWB);
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
                ay += RROWB;                                                                                  \
                const float yo = osig_of(y, hb);                                                              \
                const v2f un = cmul2(mk2(y, yo), rho);                    /* rho_k y_k, normalised next step */ \
                u = un.x;                                                                                     \
                const int kn = kk_ + 1 < CH2 ? kk_ + 1 : 0;               /* chunk end: a dummy, retired below */ \
                bcast_issue_tab(aUw, aUr, u, aRho + kn * 256, qu, rho);                                       \
                FORM_M(rdlane(sv, kn))                                    /* in the shadow of the broadcast */ \
                xsq = y * y;                                                                                  \
                write_lane(nvec, nprev, (kk_ - 1) & (CH2 - 1));                                               \
            }
            // a full chunk runs unrolled with compile-time step numbers: every LDS offset is an immediate, the lane selects of
            // v_readlane / v_writelane are constants (no M0 save / restore), no loop bookkeeping on the serial chain
#define CHAIN_STEPC(KK)                                                                                       \
            {                                                                                                 \
                lds_wait_lo<5>(qu);                                                                           \
                v2f am;                                                                                       \
                float nprev;                                                                                  \
                mv1r_lo(MM, qu, am, xsq);                                                                     \
                lds_wait_hi_t_after<0>(qu, rho, am);                                                          \
                mv1r_hi(MM, qu, am, xsq, nprev);                                                              \
                const float inv = __builtin_amdgcn_rsqf(vmax_s(nprev, 1e-12f));                               \
                const float y = inv * (u + swapadd(am.x, am.y));                                              \
                lds_write32_imm<(KK) * RROWB>(ay, y);                                                         \
                const float yo = osig_of(y, hb);                                                              \
                const v2f un = cmul2(mk2(y, yo), rho);                                                        \
                u = un.x;                                                                                     \
                bcast_issue_tab_off<(((KK) + 1) & (CH2 - 1)) * 256>(aUw, aUr, u, aRho, qu, rho);              \
                FORM_M(rdlane(sv, ((KK) + 1) & (CH2 - 1)))                                                    \
                xsq = y * y;                                                                                  \
                write_lane_c<((KK) + CH2 - 1) & (CH2 - 1)>(nvec, nprev);                                      \
            }
            if (cnt == CH2) {
                CHAIN_STEPC(0)
                if (SAVE && c > 0 && lane < CH2) sc[(size_t)((c - 1) >> 1) * 128 + ((c - 1) & 1) * CH2 + lane] = nvec;
                CHAIN_STEPC(1) CHAIN_STEPC(2) CHAIN_STEPC(3) CHAIN_STEPC(4) CHAIN_STEPC(5) CHAIN_STEPC(6) CHAIN_STEPC(7)
                stage_commit<4>(nxt, lane, sr);                        // (requested eight steps ago)
                stage_load<4>(rho4, cn * CH2 + 16, N, lane, sr);       // rows 16 .. 31
                CHAIN_STEPC(8) CHAIN_STEPC(9) CHAIN_STEPC(10) CHAIN_STEPC(11) CHAIN_STEPC(12) CHAIN_STEPC(13) CHAIN_STEPC(14)
                CHAIN_STEPC(15) CHAIN_STEPC(16) CHAIN_STEPC(17) CHAIN_STEPC(18) CHAIN_STEPC(19) CHAIN_STEPC(20) CHAIN_STEPC(21)
                CHAIN_STEPC(22) CHAIN_STEPC(23)
                stage_commit<4>(nxt + 4 * 64, lane, sr);
                CHAIN_STEPC(24) CHAIN_STEPC(25) CHAIN_STEPC(26) CHAIN_STEPC(27) CHAIN_STEPC(28)
                CHAIN_STEPC(29) CHAIN_STEPC(30) CHAIN_STEPC(31)
            } else {                                                   // the clip's last, partial chunk (nothing follows it)
                CHAIN_STEP(0)
                if (SAVE && c > 0 && lane < CH2) sc[(size_t)((c - 1) >> 1) * 128 + ((c - 1) & 1) * CH2 + lane] = nvec;
                for (int kk = 1; kk < cnt; ++kk) CHAIN_STEP(kk)
            }
#undef CHAIN_STEPC
#undef CHAIN_STEP
            lds_wait_hi_t<0>(qu, rho);                                 // everything of this chunk has landed
            flag_store(aProd, c + 1, lane);                            // publish (ordered behind the chunk's y rows)
            if (c + 1 < NC2) {
                aRho = lds_addr(&stR[w][(c + 1) & 1][0]) + i * 8;     // both halves of the next chunk are in place
                sv = LEGACY ? P.dt * (xa1 - xa0) : (xa1 - xa0) / A;    // the next chunk's s_k
                FORM_M(rdlane(sv, 0))                                  // the last in-loop FORM_M used the stale lane 0
                bcast_issue_tab_sync(aUw, aUr, u, aRho, qu, rho);      // (its registers cross the back-edge: the wait sits inside)
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
    // exact three-way bf16 split of two floats (even element in the low half of every packed word)
    auto split3 = [](float fe, float fo, unsigned& H, unsigned& M, unsigned& L) {
        const unsigned xe = __float_as_uint(fe), xo = __float_as_uint(fo);
        H = __builtin_amdgcn_perm(xo, xe, 0x07060302u);
        const float re = fe - __uint_as_float(xe & 0xFFFF0000u), ro = fo - __uint_as_float(xo & 0xFFFF0000u);
        const unsigned me = __float_as_uint(re), mo = __float_as_uint(ro);
        M = __builtin_amdgcn_perm(mo, me, 0x07060302u);
        const float le = re - __uint_as_float(me & 0xFFFF0000u), lo = ro - __uint_as_float(mo & 0xFFFF0000u);
        L = __builtin_amdgcn_perm(__float_as_uint(lo), __float_as_uint(le), 0x07060302u);   // <= 8 bits left: exact
    };
    auto frag = [](const unsigned (&f)[4]) { return __builtin_bit_cast(bf8, v4u{f[0], f[1], f[2], f[3]}); };
    // B operand: W[m][n], the real 64 x 64 form of H = R + R^dagger acting on (re, im)-interleaved vectors,
    //   (H y)[n = 2 i + c] = sum_m y[m = 2 j + c'] W[m][n]:  W = Hr_ij for c' = c,  -Hi_ij for (c', c) = (1, 0),  +Hi_ij for (0, 1).
    // Lane (col = lane & 31, hk = lane >> 5) holds W[16 s + 8 hk + e][32 t + col], e = 0..7, for tile t and k-step s.
    typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
    typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
    auto split2h = [](float fe, float fo, unsigned& H, unsigned& L) {           // two fp16 pieces, round to nearest (11 + 1 + 11 + 1 bits)
        const hf2 hh = {(_Float16)fe, (_Float16)fo};                            // plain casts: hipcc must see who produces an MFMA operand
        const hf2 ll = {(_Float16)(fe - (float)hh.x), (_Float16)(fo - (float)hh.y)};
        H = __builtin_bit_cast(unsigned, hh);
        L = __builtin_bit_cast(unsigned, ll);
    };
    // the same for the pair p * sc (sc a power of two): ONE packed multiply, and each residual as ONE v_fma_mix_f32 (p sc - hi, the fp16
    // half read in place); hipcc makes a v_cvt_f32_f16 + v_fma_f32 of the C form.  The asm results feed compiler-visible conversions, never
    // an MFMA directly, so the hazard of DESIGN 4.3e cannot arise.
    auto split2h_scaled = [](v2f p, float sc, unsigned& H, unsigned& L) {
        const v2f t = p * mk2(sc, sc);
        const hf2 hh = {(_Float16)t.x, (_Float16)t.y};
        H = __builtin_bit_cast(unsigned, hh);
        float re, ro;
        asm("v_fma_mix_f32 %0, %2, %4, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
            "v_fma_mix_f32 %1, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(re), "=&v"(ro) : "v"(p.x), "v"(p.y), "v"(sc), "v"(H));
        const hf2 ll = {(_Float16)re, (_Float16)ro};
        L = __builtin_bit_cast(unsigned, ll);
    };
    auto fragh = [](const unsigned (&f)[4]) { return __builtin_bit_cast(hf8, v4u{f[0], f[1], f[2], f[3]}); };
    unsigned WH[2][4][4], WM[HF16 ? 1 : 2][HF16 ? 1 : 4][4], WL[2][4][4];
    float sW = 1.f, Qn = 0.f, Rn = 0.f;              // HF16: scale of W; Frobenius norms of Q and R (the bound of |y_k|)
    {
        const int col = lane & 31, hk = lane >> 5;
        if constexpr (HF16) {
            float mw = 0.f, q2 = 0.f, r2 = 0.f;
            for (int idx = lane; idx < DPW * DPW; idx += 64) {
                const float2 r = P.R[idx], rt = P.RT[idx], q = P.Q[idx];
                mw = fmaxf(mw, fmaxf(fabsf(r.x + rt.x), fabsf(r.y - rt.y)));
                q2 += q.x * q.x + q.y * q.y;
                r2 += r.x * r.x + r.y * r.y;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) mw = fmaxf(mw, __shfl_xor(mw, off, 64));
            Qn = 1.001f * sqrtf(sum64(q2));
            Rn = 1.001f * sqrtf(sum64(r2));
            // the largest power of two with max |W| sW < 2^15 (W = 0: any scale)
            int se = 15 - ((int)((__float_as_uint(fmaxf(mw, 1e-30f)) >> 23) & 0xFFu) - 126);
            se = se > 60 ? 60 : se < -60 ? -60 : se;
            sW = __uint_as_float((unsigned)(127 + se) << 23);
            sW = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(sW)));
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int n = 32 * t + col, ii = n >> 1, cc = n & 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                float wv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int m = 16 * ks + 8 * hk + e, jj = m >> 1, cp = m & 1;
                    const float2 r = P.R[ii * DPW + jj], rt = P.RT[ii * DPW + jj];      // R[ii][jj], R[jj][ii]
                    const float hr = r.x + rt.x, hi = r.y - rt.y;                         // (R + R^dagger)[ii][jj]
                    wv[e] = cp == cc ? hr : (cc ? hi : -hi);
                }
#pragma unroll
                for (int e2 = 0; e2 < 4; ++e2) {
                    if constexpr (HF16) split2h(wv[2 * e2] * sW, wv[2 * e2 + 1] * sW, WH[t][ks][e2], WL[t][ks][e2]);
                    else split3(wv[2 * e2], wv[2 * e2 + 1], WH[t][ks][e2], WM[t][ks][e2], WL[t][ks][e2]);
                }
            }
        }
    }
    // A operand: lane (row = lane & 31 = step of the chunk, hk) reads y_k[16 s + 8 hk + e] from the ring;
    // C/D layout of a tile: column n = 32 t + (lane & 31), rows (steps) (r & 3) + 8 (r >> 2) + 4 (lane >> 5), r = 0..15
    const int crow = lane & 31, chk = lane >> 5;
    const float* ringw = &ring[w][0][0];
    float* pew = &pe[w][0];
    float2* st = SAVE ? reinterpret_cast<float2*>(P.hst + (size_t)b * N * 128) : nullptr;   // rows of 64 (y[n], (H y)[n]) pairs
    float loss = 0.f;
    float f_below = 2.0f * P.R[0].x, n_below = 1.f;                    // LEGACY: y^dagger H y and |y|^2 of the step below the chunk (psi_0 = e_0)
    for (int c = 0; c < NC2; ++c) {
        const int kbeg = c * CH2;
        const int cnt = (N - kbeg) < CH2 ? (N - kbeg) : CH2;
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f;
        const float x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        if (!DIAG_NO_CHAIN)
            while (flag_load(aProd) < c + 1) __builtin_amdgcn_s_sleep(8);     // a chunk takes ~20000 cycles: poll rarely
        const float* rh = ringw + (c & 1) * (CH2 * RLD);               // this chunk's 32 rows (rows >= cnt: stale, ignored below)
        v16f acc0 = {}, acc1 = {};
        float sY = 1.f, unsc = 1.f;
        if constexpr (HF16) {
            // the chunk's rows: |y_k|_inf <= |y_k|_2 <= 1 + |Q|_F + |s_k| |R|_F (inv_{k-1} |ut| <= 1); s_k of this chunk sits one step per lane
            float smax = fabsf((x1 - x0) / A);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) smax = fmaxf(smax, __shfl_xor(smax, off, 64));
            const float bnd = 1.001f * (1.0f + Qn + smax * Rn);
            int se = 14 - ((int)((__float_as_uint(bnd) >> 23) & 0xFFu) - 126);
            se = se > 60 ? 60 : se < -60 ? -60 : se;
            sY = __uint_as_float(__builtin_amdgcn_readfirstlane((unsigned)(127 + se) << 23));
            unsc = __uint_as_float(0x7F000000u - __float_as_uint(sY)) * __uint_as_float(0x7F000000u - __float_as_uint(sW));   // exact 1 / (sY sW)
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float4 f0 = *reinterpret_cast<const float4*>(rh + crow * RLD + 16 * ks + 8 * chk);
            const float4 f1 = *reinterpret_cast<const float4*>(rh + crow * RLD + 16 * ks + 8 * chk + 4);
            if constexpr (HF16) {
                unsigned AH[4], AL[4];
                split2h_scaled(mk2(f0.x, f0.y), sY, AH[0], AL[0]);
                split2h_scaled(mk2(f0.z, f0.w), sY, AH[1], AL[1]);
                split2h_scaled(mk2(f1.x, f1.y), sY, AH[2], AL[2]);
                split2h_scaled(mk2(f1.z, f1.w), sY, AH[3], AL[3]);
#define MF3(ACC, T_)                                                                                           \
                ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fragh(AH), fragh(WH[T_][ks]), ACC, 0, 0, 0);      \
                ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fragh(AH), fragh(WL[T_][ks]), ACC, 0, 0, 0);      \
                ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(fragh(AL), fragh(WH[T_][ks]), ACC, 0, 0, 0);
                MF3(acc0, 0)
                MF3(acc1, 1)
#undef MF3
            } else {
            unsigned AH[4], AM[4], AL[4];
            split3(f0.x, f0.y, AH[0], AM[0], AL[0]);
            split3(f0.z, f0.w, AH[1], AM[1], AL[1]);
            split3(f1.x, f1.y, AH[2], AM[2], AL[2]);
            split3(f1.z, f1.w, AH[3], AM[3], AL[3]);
#define MF6(ACC, T_)                                                                                           \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(AH), frag(WH[T_][ks]), ACC, 0, 0, 0);           \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(AH), frag(WM[HF16 ? 0 : T_][HF16 ? 0 : ks]), ACC, 0, 0, 0);           \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(AM), frag(WH[T_][ks]), ACC, 0, 0, 0);           \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(AH), frag(WL[T_][ks]), ACC, 0, 0, 0);           \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(AL), frag(WH[T_][ks]), ACC, 0, 0, 0);           \
            ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(AM), frag(WM[HF16 ? 0 : T_][HF16 ? 0 : ks]), ACC, 0, 0, 0);
            MF6(acc0, 0)
            MF6(acc1, 1)
#undef MF6
            }
        }
        if constexpr (HF16) {                                          // packed: one v_pk_mul_f32 per two accumulator registers
            const v2f u2 = mk2(unsc, unsc);
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const v2f a = mk2(acc0[r], acc0[r + 1]) * u2, b2 = mk2(acc1[r], acc1[r + 1]) * u2;
                acc0[r] = a.x; acc0[r + 1] = a.y; acc1[r] = b2.x; acc1[r + 1] = b2.y;
            }
        }
        // y_k[n] in the C/D layout, products y (H y), stash rows
        float yc0[16], yc1[16], pr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int stp = (r & 3) + 8 * (r >> 2) + 4 * chk;
            yc0[r] = rh[stp * RLD + crow];
            yc1[r] = rh[stp * RLD + 32 + crow];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) pr[r] = yc0[r] * acc0[r] + yc1[r] * acc1[r];
        float pn[LEGACY ? 16 : 1];
        if constexpr (LEGACY) {
#pragma unroll
            for (int r = 0; r < 16; ++r) pn[r] = yc0[r] * yc0[r] + yc1[r] * yc1[r];
        }
        if (SAVE) {
            // Stash row (kbeg + step) = 64 pairs (y[n], (H y)[n]) = 512 contiguous bytes.  Register r holds step s0 = (r & 3) +
            // 8 (r >> 2) in lanes 0-31 and step s0 + 4 in lanes 32-63, for n = 0..31 (tile 0) and n = 32..63 (tile 1): one
            // v_permlane32_swap between the two tiles puts ALL of row s0 into one register and all of row s0 + 4 into the
            // other, so every store instruction writes one whole row (lane = n).
            // A full chunk (all but a clip's last): address = uniform base (SGPR pair, one per eight rows: the immediate reaches 4095) + this
            // lane's 8 bytes (one VGPR for the whole kernel) + the row's immediate -- no 64-bit VALU address arithmetic and no per-row test.
            // (Stores by asm: dwordx2 data has no write-after-store hazard; nothing in this kernel reads the rows back.)
            char* cbase = reinterpret_cast<char*>(st + (size_t)kbeg * 64);
            const unsigned loff = (unsigned)lane * 8u;
            if (cnt == CH2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int s0 = (r & 3) + 8 * (r >> 2);
                    const auto ys = __builtin_amdgcn_permlane32_swap(__float_as_uint(yc0[r]), __float_as_uint(yc1[r]), false, false);
                    const auto hs = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[r]), __float_as_uint(acc1[r]), false, false);
                    const v2f lo = mk2(__uint_as_float(ys[0]), __uint_as_float(hs[0])), hi = mk2(__uint_as_float(ys[1]), __uint_as_float(hs[1]));
                    const char* q0 = cbase + (s0 >> 3) * 4096;              // rows 8 q .. 8 q + 7
                    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" :: "v"(loff), "v"(lo), "s"(q0), "n"((s0 & 7) * 512) : "memory");
                    asm volatile("global_store_dwordx2 %0, %1, %2 offset:%3" :: "v"(loff), "v"(hi), "s"(q0), "n"(((s0 + 4) & 7) * 512) : "memory");
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int s0 = (r & 3) + 8 * (r >> 2);
                    const auto ys = __builtin_amdgcn_permlane32_swap(__float_as_uint(yc0[r]), __float_as_uint(yc1[r]), false, false);
                    const auto hs = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc0[r]), __float_as_uint(acc1[r]), false, false);
                    if (s0 < cnt)
                        *reinterpret_cast<float2*>(cbase + s0 * 512 + loff) = make_float2(__uint_as_float(ys[0]), __uint_as_float(hs[0]));
                    if (s0 + 4 < cnt)
                        *reinterpret_cast<float2*>(cbase + (s0 + 4) * 512 + loff) = make_float2(__uint_as_float(ys[1]), __uint_as_float(hs[1]));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flag_store(aCons, c + 1, lane);                                // every ring read has landed: the half is free
        // e_k = sum_n y_k[n] (H y_k)[n]: transpose the per-column products through LDS, then lane (k, hh) sums columns
        // 16 hh .. 16 hh + 15 of row k and the two halves are added
#pragma unroll
        for (int r = 0; r < 16; ++r) pew[((r & 3) + 8 * (r >> 2) + 4 * chk) * PE2_LD + crow] = pr[r];
        __builtin_amdgcn_wave_barrier();
        float evec;
        {
            const float4* rowp = reinterpret_cast<const float4*>(pew + crow * PE2_LD + 16 * chk);
            const float4 q0 = rowp[0], q1 = rowp[1], q2 = rowp[2], q3 = rowp[3];
            const float part = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w)) +
                               (((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w)));
            evec = swapadd(part, part);
        }
        __builtin_amdgcn_wave_barrier();
        const float incv = x1 - x0;
        float lv;
        if constexpr (LEGACY) {
            // |y_k|^2 the same way, then the expectation on the normalised state of the step below
            float nvec2;
#pragma unroll
            for (int r = 0; r < 16; ++r) pew[((r & 3) + 8 * (r >> 2) + 4 * chk) * PE2_LD + crow] = pn[r];
            __builtin_amdgcn_wave_barrier();
            {
                const float4* rowp = reinterpret_cast<const float4*>(pew + crow * PE2_LD + 16 * chk);
                const float4 q0 = rowp[0], q1 = rowp[1], q2 = rowp[2], q3 = rowp[3];
                const float part = ((q0.x + q0.y) + (q0.z + q0.w)) + ((q1.x + q1.y) + (q1.z + q1.w)) +
                                   (((q2.x + q2.y) + (q2.z + q2.w)) + ((q3.x + q3.y) + (q3.z + q3.w)));
                nvec2 = swapadd(part, part);
            }
            __builtin_amdgcn_wave_barrier();
            float fb = __shfl_up(evec, 1, 64), nb = __shfl_up(nvec2, 1, 64);       // lanes k and k + 32 both hold step k
            if (crow == 0) { fb = f_below; nb = n_below; }
            f_below = rdlane(evec, CH2 - 1);
            n_below = rdlane(nvec2, CH2 - 1);
            const float invb = 1.0f / sqrtf(fmaxf(nb, 1e-12f));       // graph.pbtxt:14350-14594
            evec = (fb * invb) * invb;                                 // e_k = psi_k^dagger (R + R^T) psi_k  (:11857-12661)
            const float d = incv - evec;
            lv = d * d / 2.0f;                                         // :12685-12819
        } else {
            const float z = (evec * incv) / A;                         // model.py:294 operation order
            lv = -logf(1.0f + z);
        }
        if (cnt == CH2) {                                              // model.py:279: sequential in time.  A full chunk unrolled: the lane
#pragma unroll                                                         // selects are immediates (v_readlane + v_add per step; the counted loop
            for (int j = 0; j < CH2; ++j) loss += rdlane(lv, j);       // below costs eight instructions per step)
        } else {
            for (int j = 0; j < cnt; ++j) loss += rdlane(lv, j);
        }
        if (SAVE && lane < CH2) sc[(size_t)(c >> 1) * 128 + 64 + (c & 1) * CH2 + lane] = evec;
    }
    if (lane == 0) loss_out[b] = loss;
}

hipError_t launch_fwd_wave2(const Dev& P, const float* audio, float* loss, bool save, bool hf16, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (save && hf16)
        hipLaunchKernelGGL((k_fwd_wave2<true, false, true>), dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    else if (save)
        hipLaunchKernelGGL((k_fwd_wave2<true, false, false>), dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    else if (hf16)
        hipLaunchKernelGGL((k_fwd_wave2<false, false, true>), dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    else
        hipLaunchKernelGGL((k_fwd_wave2<false, false, false>), dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    return hipGetLastError();
}

hipError_t launch_fwd_legacy_wave(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (save)
        hipLaunchKernelGGL((k_fwd_wave2<true, true>), dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    else
        hipLaunchKernelGGL((k_fwd_wave2<false, true>), dim3(nb), dim3(128 * WAVES), 0, s, P, audio, loss);
    return hipGetLastError();
}

}  // namespace cmps
