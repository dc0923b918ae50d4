// Wave-per-clip kernels for bond dimensions D <= 16 (BASELINE configs[1]: D=16, T=4096, B=256; the reference's default is
// bond_dim = 8, train.py:41).  Same arithmetic, same stash / scalar / slab formats as the D <= 32 kernels (cmps_wave2.hip,
// cmps_wave.hip), in a lane layout that multiplies no padding:
//
//   lane l = (i = l & 15, q = l >> 4): row i of every matrix, K-quarter q (columns 4q .. 4q+3, four (re, im) pairs = 8 VGPRs
//   per matrix instead of 32).  SPLIT16 layout of a complex 16-vector X: one float per lane, x = (q & 2) ? Im X_i : Re X_i,
//   every value held twice (q & 1 = 0, 1); its partner component sits 32 lanes away, exactly as in the 32-row layout, so
//   osig_of / cmul2 / the rotation arithmetic are shared.  A mat-vec is 8 packed FMAs per lane; the four K-quarters are
//   combined by one v_permlane32_swap + add (lands Re in lanes 0-31, Im in lanes 32-63) and one v_permlane16_swap + add.
//   Sums over all 64 lanes count every component twice: wave reductions carry a factor 1/2 (exact).
//
// At B = 256 there is one clip per CU, so the work nothing waits for runs in a second wave on ANOTHER SIMD: a workgroup is
// one clip = two waves.  Forward: chain wave (y_k = u + M_k u, M_k = Q + s_k R, norm, rotation) + loss wave (H y, e_k, stash,
// loss), hand-over through a ring in LDS per 32-step chunk.  Reverse: chain wave (cotangent recursion, one mat-vec with
// M_k = Q + s_k R^dagger) + gradient wave (the three rank-1 updates per step as EXACT fp32 v_mfma_f32_16x16x4_f32: K = 4 is
// {re, im} x two steps, the split16 layout is the operand layout), hand-over per octet of steps.  Two LDS counters per
// clip, no barrier (LDS operations of one wave complete in order).
#include "cmps_wave_util.h"

namespace cmps {

namespace {

constexpr int D16 = 16;            // rows of this layout
constexpr int SD = 32;             // row stride (entries) of the matrices / rho / slabs in the workspace: DP = padded_D(D) = 32
constexpr int CH16 = 32;           // steps per chunk (ring half, rho staging, per-chunk scalar math) in the forward
constexpr int RING16 = 2 * CH16;   // forward ring slots
constexpr int PE16_LD = 33;        // row stride of the loss wave's product buffer
constexpr int CS16 = 8;            // reverse: steps per staged chunk of stash / rho rows = one octet
constexpr int BROW = 256;          // reverse ring: bytes of one row (64 floats)
constexpr int BSLOT = 3 * BROW;    // ybar | yhat | u_k of one step
constexpr int BHALF = 8 * BSLOT;   // one octet

typedef float v4acc __attribute__((ext_vector_type(4)));

#ifndef POLL_SLEEP
#define POLL_SLEEP 1
#endif

__device__ __forceinline__ int flag_load16(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void flag_store16(unsigned addr, int v, int lane) {
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

// ---- 16-row mat-vec: this quarter's four columns, one chain of 8 packed FMAs ----
__device__ __forceinline__ v2f mv16(const v2f (&M)[4], const v4f (&q)[2]) {
    v2f acc;
    asm(CM_FIRST(0, 1, 5) CM(0, 2, 6) CM(0, 3, 7) CM(0, 4, 8)
        : "=&v"(acc)
        : "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(lo2(q[0])), "v"(hi2(q[0])), "v"(lo2(q[1])), "v"(hi2(q[1])));
    return acc;
}
// the same with the wave reduction of `x` threaded through it; `tot` (SGPR) = sum of x over the 64 lanes
__device__ __forceinline__ v2f mv16r(const v2f (&M)[4], const v4f (&q)[2], float& x, float& tot) {
    v2f acc;
    // a DPP step reads what the previous one wrote: two wait states, supplied by the two packed FMAs in between
    asm(CM_FIRST(0, 3, 7)
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        CM(0, 4, 8)
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        CM(0, 5, 9)
        "v_add_f32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        CM(0, 6, 10)
        "v_add_f32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_readlane_b32 %2, %1, 63"
        : "=&v"(acc), "+v"(x), "=s"(tot)
        : "v"(M[0]), "v"(M[1]), "v"(M[2]), "v"(M[3]), "v"(lo2(q[0])), "v"(hi2(q[0])), "v"(lo2(q[1])), "v"(hi2(q[1])));
    return acc;
}
// (partial.x, partial.y) of the four K-quarters -> split16 total in every lane
__device__ __forceinline__ float combine16(float px, float py) {
    const float s1 = swapadd_after_asm(px, py);      // lanes 0-31: Re over quarters (q, q+2); lanes 32-63: Im
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(s1), __float_as_uint(s1), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);      // + the neighbouring 16-lane row
}

// ---- LDS traffic (hidden from hipcc's waitcnt bookkeeping; outputs valid after a matching wait) ----
// broadcast: every lane writes its split16 value (both holders of a component write the same word); this quarter then reads its
// four complex entries and one 8-byte table entry
__device__ __forceinline__ void bcast16_tab(unsigned wr, unsigned rd, float mine, unsigned tab, v4f (&o)[2], v2f& t) {
    asm volatile("ds_write_b32 %3, %4\n\tds_read_b128 %0, %5\n\tds_read_b128 %1, %5 offset:16\n\tds_read_b64 %2, %6"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(t) : "v"(wr), "v"(mine), "v"(rd), "v"(tab) : "memory");
}
template <int N>
__device__ __forceinline__ void wait16_t(v4f (&o)[2], v2f& t) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(o[0]), "+v"(o[1]), "+v"(t) : "n"(N) : "memory");
}
__device__ __forceinline__ void rows16_own(unsigned rd, unsigned own, v4f (&o)[2], float& mine) {
    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %3 offset:16\n\tds_read_b32 %2, %4"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(mine) : "v"(rd), "v"(own) : "memory");
}
template <int N>
__device__ __forceinline__ void wait16_own(v4f (&o)[2], float& mine) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(o[0]), "+v"(o[1]), "+v"(mine) : "n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ring16_bcast(unsigned wr, unsigned rd, float mine, v4f (&o)[2]) {
    asm volatile("ds_write_b32 %2, %3 offset:%5\n\tds_read_b128 %0, %4 offset:%5\n\tds_read_b128 %1, %4 offset:%6"
                 : "=&v"(o[0]), "=&v"(o[1]) : "v"(wr), "v"(mine), "v"(rd), "n"(OFF), "n"(OFF + 16) : "memory");
}
template <int N>
__device__ __forceinline__ void wait16(v4f (&o)[2]) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(o[0]), "+v"(o[1]) : "n"(N) : "memory");
}
template <int OFF>
__device__ __forceinline__ void write16_off(unsigned wr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(wr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void ring16_read3(unsigned rd0, unsigned rd, float& yb, float& yh, float& uk) {
    asm volatile("ds_read_b32 %0, %3 offset:%5\n\tds_read_b32 %1, %4 offset:%6\n\tds_read_b32 %2, %4 offset:%7"
                 : "=&v"(yb), "=&v"(yh), "=&v"(uk) : "v"(rd0), "v"(rd), "n"(OFF), "n"(OFF + BROW), "n"(OFF + 2 * BROW) : "memory");
}
template <int N>
__device__ __forceinline__ void wait16_3(float& a, float& b, float& c) {
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N) : "memory");
}

// chunk staging of rho rows: only the first 16 entries (128 B) of each 256-B row; NQ * 8 rows
template <int NQ>
__device__ __forceinline__ void stage16_load(const float4* __restrict__ tab, int row0, int max_row, int lane, v4f (&r)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int e = q * 64 + lane;
        int row = row0 + (e >> 3);
        row = row < max_row ? row : max_row;
        const float4 t = tab[(size_t)row * 16 + (e & 7)];
        r[q] = v4f{t.x, t.y, t.z, t.w};
    }
}

// In-kernel stamps of the serial chains (diagnostic builds only: -DCMPS_DIAG -DW16_TIMING, scripts/stamps_c2.py): s_memtime at
// the phase boundaries of a chain step, pinned behind the value that ends the phase; block 0 prints the averages.
#if defined(CMPS_DIAG) && defined(W16_TIMING)
#define W16_STAMP(T_, DEP_) { asm volatile("" : "+v"(DEP_)); T_ = __builtin_readcyclecounter(); }
#define W16_ON 1
#else
#define W16_STAMP(T_, DEP_)
#define W16_ON 0
#endif

struct Pre16 {
    float yh, yho, yhp, un, uno, pre;
    v2f rho;
    float inv, s, dtk, rad;
};

}  // namespace

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
template <bool SAVE>
__global__ __launch_bounds__(128, 1) void k_fwd_wave16(Dev P, const float* __restrict__ audio, float* __restrict__ loss_out) {
    __shared__ __attribute__((aligned(16))) float4 stR[CH16 * 8];          // rho rows (16 entries) of the chain wave's chunk
    __shared__ __attribute__((aligned(16))) float2 bcU[D16];
    __shared__ __attribute__((aligned(16))) float2 ring[RING16][D16];      // y_k, interleaved (re, im) per component
    __shared__ float pe[64 * PE16_LD];
    __shared__ int flags[2];                                                // 0: prod, 1: cons
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4, hq = q >> 1;
    const bool hb = hq != 0;
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x;
    const int N = P.N, T = P.T, NC2 = (N + CH16 - 1) / CH16, NC = (N + CH - 1) / CH;
    const float* xrow = audio + (size_t)b * T;
    const float A = dev_A(P);
    float* sc = SAVE ? P.scal + scal_off(b, NC, 0) : nullptr;
    const unsigned aRing = lds_addr(&ring[0][0]);
    const unsigned aProd = lds_addr(&flags[0]), aCons = lds_addr(&flags[1]);

    if (role == 0) {
        // ------------------------------------------------------------------ chain wave
        v2f MR[4], MQ[4], MM[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            MR[m] = ld2(&P.R[i * SD + 4 * q + m]);
            MQ[m] = ld2(&P.Q[i * SD + 4 * q + m]);
        }
        const unsigned aUw = lds_addr(&bcU[0]) + i * 8 + hq * 4, aUr = lds_addr(&bcU[0]) + q * 32;
        const unsigned aYw = aRing + i * 8 + hq * 4;
        const unsigned aRho = lds_addr(&stR[0]) + i * 8;
        const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
        v4f sr[4];
        stage16_load<4>(rho4, 0, N, lane, sr);
        float xa0 = lane < T ? xrow[lane] : 0.f;
        float xa1 = lane + 1 < T ? xrow[lane + 1] : 0.f;
        stage_commit<4>(stR, lane, sr);
        const float2 p0 = P.psi0[i];
        float u = hb ? p0.y : p0.x;
        v4f qu[2];
        v2f rho;
        float xsq = lane == 0 ? 2.f : 0.f;      // "2 |y_{-1}|^2" = 2: psi_0 arrives normalised (sums count components twice)
        float nvec = 1.f;
        float sv = (xa1 - xa0) / A;             // model.py:263, 303
#define FORM_M16(S_)                                                                                          \
        {                                                                                                     \
            const float s_ = (S_);                                                                            \
            const v2f s2_ = mk2(s_, s_);                                                                      \
            _Pragma("unroll") for (int m = 0; m < 4; ++m) MM[m] = __builtin_elementwise_fma(MR[m], s2_, MQ[m]); \
        }
        FORM_M16(rdlane(sv, 0))
        bcast16_tab(aUw, aUr, u, aRho, qu, rho);
#if W16_ON
        unsigned long long tA = 0, tB = 0, tC = 0, accWait = 0, accMv = 0, accTail = 0, accN = 0;
        const unsigned long long tStart = __builtin_readcyclecounter();
#endif
        for (int c = 0; c < NC2; ++c) {
            const int kbeg = c * CH16;
            const int cnt = (N - kbeg) < CH16 ? (N - kbeg) : CH16;
            {
                const int cn = c + 1 < NC2 ? c + 1 : NC2 - 1;
                stage16_load<4>(rho4, cn * CH16, N, lane, sr);
                const int idx = cn * CH16 + lane;
                xa0 = idx < T ? xrow[idx] : 0.f;
                xa1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            }
            if (c >= 2)                                               // the ring half about to be overwritten
                while (flag_load16(aCons) < c - 1) __builtin_amdgcn_s_sleep(POLL_SLEEP);
            unsigned ay = aYw + (c & 1) * (CH16 * 128);
            for (int kk = 0; kk < cnt; ++kk) {
                wait16_t<0>(qu, rho);
                W16_STAMP(tA, qu[1])
#if W16_ON
                if (tC) { accWait += tA - tC; ++accN; }
#endif
                float tot;
                const v2f am = mv16r(MM, qu, xsq, tot);                // tot = 2 |y_{k-1}|^2
                const float nprev = 0.5f * tot;
                const float inv = __builtin_amdgcn_rsqf(fmaxf(nprev, 1e-12f));     // model.py:332
                float y = inv * (u + combine16(am.x, am.y));
                W16_STAMP(tB, y)
                lds_write32(ay, y);
                ay += 128;
                const float yo = osig_of(y, hb);
                const v2f un = cmul2(mk2(y, yo), rho);                 // rho_k y_k, normalised next step
                u = un.x;
                const int kn = kk + 1 < CH16 ? kk + 1 : 0;             // chunk end: a dummy, retired below
                bcast16_tab(aUw, aUr, u, aRho + kn * 128, qu, rho);
                FORM_M16(rdlane(sv, kn))                               // in the shadow of the broadcast
                xsq = y * y;
                if (kk > 0 || c > 0) nvec = (lane == ((kk - 1) & (CH16 - 1))) ? nprev : nvec;
                if (SAVE && kk == 0 && c > 0 && lane < CH16) {
                    sc[(size_t)((c - 1) >> 1) * 128 + ((c - 1) & 1) * CH16 + lane] = nvec;
                }
                W16_STAMP(tC, xsq)
#if W16_ON
                accMv += tB - tA; accTail += tC - tB;
#endif
            }
#if W16_ON
            tC = 0;                                                    // the chunk hand-over is not a step
#endif
            wait16_t<0>(qu, rho);                                      // everything of this chunk has landed
            flag_store16(aProd, c + 1, lane);                          // publish (ordered behind the chunk's y rows)
            if (c + 1 < NC2) {
                stage_commit<4>(stR, lane, sr);
                sv = (xa1 - xa0) / A;
                FORM_M16(rdlane(sv, 0))
                bcast16_tab(aUw, aUr, u, aRho, qu, rho);
            }
        }
#undef FORM_M16
#if W16_ON
        if (blockIdx.x == 0 && lane == 0) {
            const unsigned long long tEnd = __builtin_readcyclecounter();
            printf("k_fwd_wave16 chain wave, cycles per step (%llu stamped steps of %d; whole loop %.1f per step incl. chunk hand-overs): "
                   "tail end -> broadcast landed (LDS write -> read round trip, exposed) %.1f | mat-vec + wave reduction + normalisation -> y %.1f | "
                   "y -> rotation, LDS write + reads issued, M_k formed %.1f\n", accN, N, (double)(tEnd - tStart) / N,
                   (double)accWait / accN, (double)accMv / accN, (double)accTail / accN);
        }
#endif
        if (SAVE) {                                                    // |y_{N-1}|^2 closes the last row
            const float nlast = 0.5f * sum64(xsq);
            const int cl = NC2 - 1;
            nvec = (lane == ((N - 1) & (CH16 - 1))) ? nlast : nvec;
            if (lane < CH16) sc[(size_t)(cl >> 1) * 128 + (cl & 1) * CH16 + lane] = nvec;
        }
        return;
    }

    // ---------------------------------------------------------------------- loss wave
    v2f MH[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const v2f r = ld2(&P.R[i * SD + 4 * q + m]);
        const v2f rt = ld2(&P.RT[i * SD + 4 * q + m]);      // R[4q+m][i]
        MH[m] = mk2(r.x + rt.x, r.y - rt.y);                // (R + R^dagger)[i][4q+m]
    }
    const unsigned aYr = aRing + q * 32, aYo = aRing + i * 8 + hq * 4;
    const unsigned aPEw = lds_addr(&pe[0]) + lane * (PE16_LD * 4);
    float2* st = SAVE ? reinterpret_cast<float2*>(P.hst + (size_t)b * N * 128) + lane : nullptr;
    float loss = 0.f;
    v4f qa[2], qb[2];
    float ya = 0.f, yb = 0.f;
    for (int c = 0; c < NC2; ++c) {
        const int kbeg = c * CH16;
        const int cnt = (N - kbeg) < CH16 ? (N - kbeg) : CH16;
        const int idx = kbeg + lane;
        const float x0 = idx < T ? xrow[idx] : 0.f;
        const float x1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        while (flag_load16(aProd) < c + 1) __builtin_amdgcn_s_sleep(POLL_SLEEP);
        const unsigned off = (c & 1) * (CH16 * 128);
        const int last = cnt - 1;
        rows16_own(aYr + off, aYo + off, qa, ya);
#define LOSS16_STEP(KK, Q, Y, QN, YN)                                                                   \
        {                                                                                              \
            const int kk_ = (KK) < last ? (KK) : last;                                                 \
            const int kn_ = (KK) + 1 < last ? (KK) + 1 : last;                                         \
            rows16_own(aYr + off + kn_ * 128, aYo + off + kn_ * 128, QN, YN);                          \
            wait16_own<3>(Q, Y);                                                                       \
            const v2f ah = mv16(MH, Q);                                                                \
            const float hs = combine16(ah.x, ah.y);                                                    \
            lds_write32(aPEw + kk_ * 4, Y * hs);                                                       \
            if (SAVE) st[(size_t)(kbeg + kk_) * 64] = make_float2(Y, hs);                              \
        }
        for (int kk = 0; kk < cnt; kk += 2) {
            LOSS16_STEP(kk, qa, ya, qb, yb)
            LOSS16_STEP(kk + 1, qb, yb, qa, ya)
        }
#undef LOSS16_STEP
        wait16_own<0>(qa, ya);
        flag_store16(aCons, c + 1, lane);                              // every ring read has landed: the half is free
        // e_k = 1/2 the sum over the 64 lanes of the stored products (every component is held twice)
        float evec;
        {
            const int ci = lane & 31, ch = lane >> 5;
            const float* col = &pe[(32 * ch) * PE16_LD + ci];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int l = 0; l < 32; l += 4) {
                a0 += col[(l + 0) * PE16_LD];
                a1 += col[(l + 1) * PE16_LD];
                a2 += col[(l + 2) * PE16_LD];
                a3 += col[(l + 3) * PE16_LD];
            }
            const float part = (a0 + a1) + (a2 + a3);
            evec = 0.5f * swapadd(part, part);
        }
        const float incv = x1 - x0;
        const float z = (evec * incv) / A;                             // model.py:294 operation order
        const float lv = -logf(1.0f + z);
        for (int j = 0; j < cnt; ++j) loss += rdlane(lv, j);           // model.py:279: sequential in time
        if (SAVE && lane < CH16) sc[(size_t)(c >> 1) * 128 + 64 + (c & 1) * CH16 + lane] = evec;
    }
    if (lane == 0) loss_out[b] = loss;
}

// ------------------------------------------------------------------------------------------------
// reverse
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128, 1) void k_bwd_wave16(Dev P, const float* __restrict__ audio) {
    __shared__ __attribute__((aligned(16))) float4 stY[CS16 * 32];         // stashed (y, H y) rows of the staged chunk
    __shared__ __attribute__((aligned(16))) float4 stR[CS16 * 16];         // rho rows (full 256-B rows)
    __shared__ __attribute__((aligned(16))) float4 scl[CH * 2];            // per-step scalars, one 32-B row per step
    __shared__ __attribute__((aligned(16))) float ring[2 * BHALF / 4];     // two octets of (ybar, yhat, u) rows
    __shared__ int flags[2];
    const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 15, q = lane >> 4, hq = q >> 1;
    const bool hb = hq != 0;
    if (threadIdx.x < 2) flags[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH;
    const int kt = N - 1, otop = kt >> 3;
    const float* xrow = audio + (size_t)b * T;
    const float* sc = P.scal + scal_off(b, NC, 0);
    const float A = dev_A(P);
    const unsigned aRing = lds_addr(&ring[0]);
    const unsigned aProd = lds_addr(&flags[0]), aCons = lds_addr(&flags[1]);
    float* slab = P.slabs + (size_t)b * P.slab_floats;
    constexpr int DD = SD * SD;

    if (role == 0) {
        // ------------------------------------------------------------------------------------------ chain wave
        v2f MRd[4], MQ[4], MM[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const v2f rt = ld2(&P.RT[i * SD + 4 * q + m]);     // R[4q+m][i]
            MRd[m] = mk2(rt.x, -rt.y);                          // R^dagger[i][4q+m]
            MQ[m] = ld2(&P.Q[i * SD + 4 * q + m]);
        }
        const unsigned aYown = lds_addr(&stY[0]) + lane * 8;
        const unsigned aRho = lds_addr(&stR[0]) + i * 8;
        const unsigned aScl = lds_addr(&scl[0]);
        const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
        const float4* sty4 = reinterpret_cast<const float4*>(P.hst + (size_t)b * N * 128);
        float facc = 0.f, accS = 0.f, accA = 0.f;
        float ra0 = 0.f, ra1 = 0.f, rdt = 0.f, rnv = 1.f, rev = 0.f;
        v4f sry[4], srr[2];
        auto scal_load = [&](int c) {
            const int idx = c * CH + lane;
            ra0 = idx < T ? xrow[idx] : 0.f;
            ra1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            rdt = P.dtk[idx];
            rnv = sc[(size_t)c * 128 + lane];
            rev = sc[(size_t)c * 128 + 64 + lane];
        };
        auto scal_commit = [&](int c) {
            const int idx = c * CH + lane;
            const float inc = ra1 - ra0;
            const float svv = inc / A;
            const float nv = rnv, ev = rev;
            const float invv = rsq_nr(fmaxf(nv, 1e-12f));
            const float invokv = nv > 1e-12f ? invv : 0.f;
            const float ex = ev * inc;                      // model.py:294 operation order
            const float z = ex / A;
            const float zbar = -1.0f / (1.0f + z);
            const float ebar = zbar * inc / A;
            const float tev = 2.0f * ebar;
            scl[2 * lane] = make_float4(svv, rdt, invv, 0.f);
            scl[2 * lane + 1] = make_float4(tev, invokv, tev * ev, 0.f);
            if (idx < N) accA += zbar * ex;
        };
        auto stage_load_all = [&](int hh) {
            stage_load512<4>(sty4, hh * CS16, N - 1, lane, sry);
            stage_load<2>(rho4, hh * CS16, N, lane, srr);
        };
        auto stage_commit_all = [&]() {
            stage_commit<4>(stY, lane, sry);
            stage_commit<2>(stR, lane, srr);
        };
        auto make_pre = [&](v2f yh2, v2f rho, v4f c0, v4f c1) -> Pre16 {
            const float yown = yh2.x, hown = yh2.y;
            Pre16 S;
            S.rho = rho;
            S.s = c0.x;
            S.dtk = c0.y;
            S.inv = c0.z;
            const float te = c1.x;
            const float invok = c1.y;
            S.rad = c1.z;
            S.pre = te * hown;
            S.yh = S.inv * yown;
            S.yhp = invok * yown;
            S.yho = osig_of(S.yh, hb);
            const v2f un = cmul2(mk2(S.yh, S.yho), rho);
            S.un = un.x;
            S.uno = un.y;
            return S;
        };
        // the phantom slots of the top octet (steps above N - 1) read as zeros in the gradient wave
        for (int sl = (kt & 7) + 1; sl < 8; ++sl) {
            const unsigned a = aRing + (otop & 1) * BHALF + sl * BSLOT + lane * 4;
            lds_write32(a, 0.f);
            lds_write32(a + BROW, 0.f);
            lds_write32(a + 2 * BROW, 0.f);
        }
        const int hl = (N - 1) / CS16;
        stage_load_all(hl);
        scal_load(hl >> 3);
        stage_commit_all();
        scal_commit(hl >> 3);
        v4f qc[2];
        v2f yh_j, rho_j;
        v4f c0_j, c1_j;
        Pre16 S;
        {
            const int jr = (N - 1) & (CS16 - 1), jc = (N - 1) & (CH - 1);
            own_issue(aYown + jr * 512, aRho + jr * 256, aScl + jc * 32, yh_j, rho_j, c0_j, c1_j);
            lds_wait_own<0>(yh_j, rho_j, c0_j, c1_j);
            S = make_pre(yh_j, rho_j, c0_j, c1_j);
        }
        float g = 0.f, go = 0.f;
        const float2 p0 = P.psi0[i];
        const float u0 = hb ? p0.y : p0.x;
        float rad_next = 0.f;
#if W16_ON
        unsigned long long bA = 0, bB = 0, bC = 0, bD = 0, bHead = 0, bShadow = 0, bMv = 0, bTail2 = 0, bN = 0;
        const unsigned long long bStart = __builtin_readcyclecounter();
#endif
        // one step of the serial chain (see cmps_wave.hip / the header of cmps_block.hip for the adjoint); ring row 0 of slot
        // KS holds ybar interleaved (re, im) per component = the broadcast source; rows 1, 2 (yhat_k, u_k) one float per lane
        auto chain_step = [&](const Pre16& S, float uk, auto have_pre, bool exact, auto kslot, unsigned aW, unsigned aR,
                              unsigned aL) -> Pre16 {
            constexpr int OFF = decltype(kslot)::value * BSLOT;
            W16_STAMP(bA, g)
#if W16_ON
            if (bD) { bTail2 += bA - bD; }
#endif
            facc += S.dtk * (go * S.un);
            const v2f yhbp = cmul2_conj_b(mk2(g, go), S.rho);              // conj(rho_k) g
            const float yhb = yhbp.x;
            float dot = rad_next;
            if (exact) {                                                   // a real branch (see cmps_wave.hip): never if-converted
                asm volatile("" ::: "memory");
                dot = 0.5f * sum64(S.yhp * yhb);
            }
            rad_next = S.rad;
            const float ybar = (yhb - dot * S.yhp) * S.inv + S.pre;
            write16_off<OFF + BROW>(aL, S.yh);                             // 1 op
            ring16_bcast<OFF>(aW, aR, ybar, qc);                           // 3 ops
            W16_STAMP(bB, MM[0])
            {   // M_k = Q + s_k R^dagger, in the shadow of the broadcast
                const v2f s2 = mk2(S.s, S.s);
#pragma unroll
                for (int m = 0; m < 4; ++m) MM[m] = __builtin_elementwise_fma(MRd[m], s2, MQ[m]);
            }
            Pre16 Sn = S;
            if constexpr (decltype(have_pre)::value) {
                lds_wait_own<4>(yh_j, rho_j, c0_j, c1_j);
                Sn = make_pre(yh_j, rho_j, c0_j, c1_j);
                uk = Sn.un;
            }
            write16_off<OFF + 2 * BROW>(aL, uk);                           // 1 op
            wait16<1>(qc);
            W16_STAMP(bC, qc[1])
            const v2f am = mv16(MM, qc);
            const float md = combine16(am.x, am.y);
            accS += md * uk;
            g = ybar + md;
            go = osig_of(g, hb);
            W16_STAMP(bD, go)
#if W16_ON
            bHead += bB - bA; bShadow += bC - bB; bMv += bD - bC; ++bN;
#endif
            return Sn;
        };
        const unsigned aWb = aRing + i * 8 + hq * 4, aRb = aRing + q * 32, aLb = aRing + lane * 4;
        for (int hh = hl; hh >= 0; --hh) {
            const int jlo = hh * CS16;
            const int jhi = (N - 2) < (jlo + CS16 - 1) ? (N - 2) : (jlo + CS16 - 1);
            stage_load_all(hh > 0 ? hh - 1 : 0);
            const bool new_scal = (hh & 7) == 0 && hh > 0;
            if (new_scal) scal_load((hh >> 3) - 1);
            const bool proj = (hh & 3) == 3 || hh == hl;   // the explicit projection: every 32 steps
            int j = jhi;
#define BWD16_STEP(KS, AW, AR, AL)                                                                            \
            {                                                                                                 \
                const int jr = j & (CS16 - 1), jc = j & (CH - 1);                                             \
                own_issue(aYown + jr * 512, aRho + jr * 256, aScl + jc * 32, yh_j, rho_j, c0_j, c1_j);        \
                S = chain_step(S, 0.f, std::true_type{}, proj && j == jhi, std::integral_constant<int, (KS)>{}, AW, AR, AL); \
                --j;                                                                                          \
            }
            if (j < jlo) {
            } else if ((j & 7) != 7) {
                const unsigned ho = (otop & 1) * BHALF;
                const unsigned aW = aWb + ho, aR = aRb + ho, aL = aLb + ho;
                switch (j & 7) {
                    case 6: BWD16_STEP(7, aW, aR, aL) [[fallthrough]];
                    case 5: BWD16_STEP(6, aW, aR, aL) [[fallthrough]];
                    case 4: BWD16_STEP(5, aW, aR, aL) [[fallthrough]];
                    case 3: BWD16_STEP(4, aW, aR, aL) [[fallthrough]];
                    case 2: BWD16_STEP(3, aW, aR, aL) [[fallthrough]];
                    case 1: BWD16_STEP(2, aW, aR, aL) [[fallthrough]];
                    default: BWD16_STEP(1, aW, aR, aL)
                }
            } else {
                const int oup = (j + 1) >> 3;
                const unsigned hu = (oup & 1) * BHALF, hd = BHALF - hu;
                {
                    const unsigned aW = aWb + hu, aR = aRb + hu, aL = aLb + hu;
                    BWD16_STEP(0, aW, aR, aL)
                }
                wait16<0>(qc);
                flag_store16(aProd, otop - oup + 1, lane);
                const int qd = otop - oup + 1;
                if (qd >= 2)
                    while (flag_load16(aCons) < qd - 1) __builtin_amdgcn_s_sleep(POLL_SLEEP);
                const unsigned aW = aWb + hd, aR = aRb + hd, aL = aLb + hd;
                BWD16_STEP(7, aW, aR, aL) BWD16_STEP(6, aW, aR, aL) BWD16_STEP(5, aW, aR, aL) BWD16_STEP(4, aW, aR, aL)
                BWD16_STEP(3, aW, aR, aL) BWD16_STEP(2, aW, aR, aL) BWD16_STEP(1, aW, aR, aL)
            }
#undef BWD16_STEP
            if (hh > 0) stage_commit_all();
            if (new_scal) scal_commit((hh >> 3) - 1);
        }
        {   // step 0: u_0 = psi_0; slot 0 of octet 0
            S = chain_step(S, u0, std::false_type{}, true, std::integral_constant<int, 0>{}, aWb, aRb, aLb);
            wait16<0>(qc);
            flag_store16(aProd, otop + 1, lane);
        }
#if W16_ON
        if (blockIdx.x == 0 && lane == 0) {
            const unsigned long long bEnd = __builtin_readcyclecounter();
            printf("k_bwd_wave16 chain wave, cycles per step (%llu steps; whole loop %.1f per step): g -> conj(rho) g, ybar, LDS write + reads "
                   "issued %.1f | M_k formed, next step's rows decoded (in the shadow of the broadcast) -> broadcast landed %.1f | mat-vec + "
                   "combine -> g %.1f | between steps (row prefetch issue, staging, octet hand-over) %.1f\n", bN,
                   (double)(bEnd - bStart) / bN, (double)bHead / bN, (double)bShadow / bN, (double)bMv / bN, (double)bTail2 / bN);
        }
#endif
        const float sumS = 0.5f * sum64(accS);                // every component is held twice
        const float sumA = sum64(accA);                       // one step per lane: no duplication
        const float ftot = swapadd(facc, facc);               // half 0: f(Re lane) + f(Im lane)
        if ((q & 1) == 0) {
            slab[4 * DD + (hb ? 2 * SD : SD) + i] = g;        // cotangent of psi_0
            if (!hb) slab[4 * DD + i] = ftot;
        }
        if (lane == 0) {
            slab[4 * DD + 3 * SD] = -(sumA / (A * A)) - sumS / A;      // k_finalize removes the Q part (Dev::abar_fix)
            slab[4 * DD + 3 * SD + 1] = 0.f;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------- gradient wave
    // Rbar += 2 ebar y y^dagger + s ybar u^dagger ;  Qbar += ybar u^dagger as exact fp32 16x16x4 MFMAs: lane (i, q) feeds
    // A[i][k = q] / B[k = q][j = i]; k = q is {re, im} (q & 2: the split16 layout) x {step k, step k-1} (q & 1: the two holders
    // of a component carry the two steps of a pair).  Re(a b^dagger): B = b; Im(a b^dagger): B = -b_osig (sign at the end).
    v4acc Rre = {}, Rim = {}, Qre = {}, Qim = {};
    {
        float ra0 = 0.f, ra1 = 0.f, rnv = 1.f, rev = 0.f, sv = 0.f, tenv = 0.f;
        auto scal_load = [&](int c) {
            const int idx = c * CH + lane;
            ra0 = idx < T ? xrow[idx] : 0.f;
            ra1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            rnv = sc[(size_t)c * 128 + lane];
            rev = sc[(size_t)c * 128 + 64 + lane];
        };
        auto scal_commit = [&](int c) {
            const int idx = c * CH + lane;
            const float inc = ra1 - ra0;
            const float z = (rev * inc) / A;
            const float zbar = -1.0f / (1.0f + z);
            const float ebar = zbar * inc / A;
            const bool live = idx < N;
            sv = live ? inc / A : 0.f;
            tenv = live ? 2.0f * ebar * rnv : 0.f;
        };
        const bool odd = (q & 1) != 0;
        int cc = kt >> 6;
        scal_load(cc);
        scal_commit(cc);
        if (cc > 0) scal_load(cc - 1);
        float yb, yh, uk, ybn, yhn, ukn;
        float pa1 = 0.f, pyb = 0.f, pa2 = 0.f, pyh = 0.f, pyho = 0.f, puk = 0.f, puko = 0.f;   // the even step of a pair
        for (int o = otop; o >= 0; --o) {
            if (((o * 8 + 7) >> 6) != cc) {
                --cc;
                scal_commit(cc);
                if (cc > 0) scal_load(cc - 1);
            }
            while (flag_load16(aProd) < otop - o + 1) __builtin_amdgcn_s_sleep(POLL_SLEEP);
            const unsigned aRd = aRing + (o & 1) * BHALF + lane * 4, aRd0 = aRing + (o & 1) * BHALF + i * 8 + hq * 4;
            const int kb = (o * 8) & 63;
            ring16_read3<7 * BSLOT>(aRd0, aRd, yb, yh, uk);
#define GRAD16_STEP(SL, A_, B_, C_, AN_, BN_, CN_)                                                            \
            {                                                                                                 \
                if constexpr ((SL) > 0) ring16_read3<((SL) > 0 ? (SL) - 1 : 0) * BSLOT>(aRd0, aRd, AN_, BN_, CN_); \
                wait16_3<((SL) > 0 ? 3 : 0)>(A_, B_, C_);                                                     \
                const float sk = rdlane(sv, kb + (SL)), tk = rdlane(tenv, kb + (SL));                         \
                const float yho = osig_of(B_, hb), uko = osig_of(C_, hb);                                     \
                const float a1 = tk * B_, a2 = sk * A_;                                                       \
                if constexpr (((SL) & 1) != 0) {                                                              \
                    pa1 = a1; pyb = A_; pa2 = a2; pyh = B_; pyho = yho; puk = C_; puko = uko;                 \
                } else {                                                                                      \
                    const float fa1 = odd ? pa1 : a1, fyb = odd ? pyb : A_, fa2 = odd ? pa2 : a2;             \
                    const float fyh = odd ? pyh : B_, fyho = odd ? pyho : yho, fuk = odd ? puk : C_, fuko = odd ? puko : uko; \
                    Rre = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1, fyh, Rre, 0, 0, 0);                       \
                    Rim = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1, fyho, Rim, 0, 0, 0);                      \
                    Qre = __builtin_amdgcn_mfma_f32_16x16x4f32(fyb, fuk, Qre, 0, 0, 0);                       \
                    Qim = __builtin_amdgcn_mfma_f32_16x16x4f32(fyb, fuko, Qim, 0, 0, 0);                      \
                    Rre = __builtin_amdgcn_mfma_f32_16x16x4f32(fa2, fuk, Rre, 0, 0, 0);                       \
                    Rim = __builtin_amdgcn_mfma_f32_16x16x4f32(fa2, fuko, Rim, 0, 0, 0);                      \
                }                                                                                             \
            }
            GRAD16_STEP(7, yb, yh, uk, ybn, yhn, ukn) GRAD16_STEP(6, ybn, yhn, ukn, yb, yh, uk)
            GRAD16_STEP(5, yb, yh, uk, ybn, yhn, ukn) GRAD16_STEP(4, ybn, yhn, ukn, yb, yh, uk)
            GRAD16_STEP(3, yb, yh, uk, ybn, yhn, ukn) GRAD16_STEP(2, ybn, yhn, ukn, yb, yh, uk)
            GRAD16_STEP(1, yb, yh, uk, ybn, yhn, ukn) GRAD16_STEP(0, ybn, yhn, ukn, yb, yh, uk)
#undef GRAD16_STEP
            flag_store16(aCons, otop - o + 1, lane);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * q + r;                            // C/D layout of the 16x16 MFMA: column = lane & 15
        const int o = row * SD + i;
        slab[o] = Rre[r];
        slab[DD + o] = -Rim[r];
        slab[2 * DD + o] = Qre[r];
        slab[3 * DD + o] = -Qim[r];
    }
}

hipError_t launch_fwd_wave16(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s) {
    if (save)
        hipLaunchKernelGGL(k_fwd_wave16<true>, dim3(P.B), dim3(128), 0, s, P, audio, loss);
    else
        hipLaunchKernelGGL(k_fwd_wave16<false>, dim3(P.B), dim3(128), 0, s, P, audio, loss);
    return hipGetLastError();
}

hipError_t launch_bwd_wave16(const Dev& P, const float* audio, hipStream_t s) {
    hipLaunchKernelGGL(k_bwd_wave16, dim3(P.B), dim3(128), 0, s, P, audio);
    return hipGetLastError();
}

}  // namespace cmps
