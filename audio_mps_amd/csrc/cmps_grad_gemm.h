// Gradient contraction of the D > 32 kernels (cmps_wide.hip: float32-faithful, cmps_pair.hip: bf16 operands):
//   Rbar = sum_{clip,k} (te_k y_k) y_k^dagger + (s_k ybar_k) u_k^dagger,   Qbar = sum ybar_k u_k^dagger      (model.py:300-325 reversed)
// as bf16 MFMA GEMMs over K = (clip, step), fp32 accumulators resident for the whole pair of clips.  A complex outer product
// C += a b^dagger is two real GEMMs over K:  Re C = [a_re | a_im] [b_re | b_im]^T,   Im C = [a_im | -a_re] [b_re | b_im]^T.
// One MFMA covers K = 16 = {re, im} x 2 clips x 4 steps: the K half (lane >> 5) is the component and a lane's eight K values are ONE
// 16-byte piece [clip][step] of an operand array op[piece or sub-unit][operand][component][row].
// Wave w (one per SIMD, 512 registers: up to 256 of them accumulators) owns the 32-row block w of Re / Im Rbar and Re / Im Qbar.  The
// five operands (te y | s ybar | ybar | y | u) of the NEXT unit are built from the float32 rows -- y from the forward's stash, ybar
// from the reverse scan, u_k = rho_{k-1} y_{k-1} / |y_{k-1}| recomputed -- by the same waves, double-buffered in LDS, one barrier per
// unit; the raw rows are fetched a unit before that.
//   NPC = 3 / 2: every operand split exactly into NPC bf16 pieces (truncation) and the piece products a + b <= NPC - 1 kept (24 / 16
//                operand bits: float32-faithful products); a unit = 4 steps, NPC (NPC + 1) / 2 groups of 6 PD / 32 MFMAs.
//   F16 (NPC = 2): the two pieces are fp16 (round to nearest: hi = f16(x), lo = f16(x - hi), |x - hi - lo| <= 2^-24 |x| inside fp16's
//                normal range) and the MFMA is v_mfma_f32_32x32x16_f16: the three products of BF16X2 with the 22 + 2 operand bits of
//                BF16X3.  fp16 has 5 exponent bits, so every operand class is scaled by a power of two per pair of clips (exact; folded
//                into the per-step scalars, so the build is no longer than BF16X2's): A operands to [.., 2^15), from max |ybar| (left
//                behind by the reverse scan, Dev::opmax) and bounds of the per-step scalars formed in a pre-pass; B operands (y, u) to
//                [.., 2^13).  Below 2^-3 of a class's scaled bound lo is subnormal (spacing 2^-24: an ABSOLUTE error <= 2^-39 of the
//                bound); the accumulators are unscaled when they are written out.
//   NPC = 1:     every operand rounded to bf16 once (the rounding points of oracle/cmps_oracle.py::psi_bf16_scan); a unit = 8 steps
//                (two sub-units = two groups), so that a barrier interval holds 12 PD / 32 MFMAs and the second sub-unit's operand
//                reads are issued behind the first one's MFMAs.
//
// The issue order of a unit is written out: slot t = MFMA t of the unit followed by a few small slices of the operand build (at most
// four VALU instructions, or one load, or the LDS stores of one operand), MFMA and slices each closed by a sched_barrier.  Measured on
// this chip (scripts/ubench/mfma_valu_overlap*.hip, profiles/r3_ubench_mfma_valu_overlap.log): behind one v_mfma_f32_32x32x16_bf16 of
// a lone wave, 1 ds_read_b128 + 3 VALU + 1/4 ds_write_b64 are free (17.9 against 16.9 ns per MFMA), 5 VALU cost +40 %, back-to-back
// dependent VALU pairs +10-20 %, v_pk_fma_f32 never hides; left to itself hipcc issues the MFMAs in runs of 20-30 and the build in runs
// of 40-150 VALU, which adds the two streams (NPC = 3 at configs[4]: MFMAs alone 11.0 ms, build alone 10.6 ms, together 17.2 ms).
// MFMA order inside a group: A piece outermost (te y | y: Re, Im; ybar | u: Re, Im; s ybar | u: Re, Im), column block innermost, so an
// A register is free after PD / 32 MFMAs and is refilled for the next group at once (single-buffered operands: 24 + 2 PD / 4
// registers), and consecutive MFMAs never share an accumulator.
#pragma once
#include <type_traits>

#include "cmps_internal.h"

namespace cmps {
namespace gg {

typedef float v4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef short bf8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

constexpr int CH = 64;       // steps per chunk of per-step scalars (the forward's scal rows)

template <int I, int E, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}
__device__ __forceinline__ unsigned pack_hi16(unsigned lo_word, unsigned hi_word) {   // (lo_word >> 16) | (hi_word & 0xFFFF0000)
    return __builtin_amdgcn_perm(hi_word, lo_word, 0x07060302u);
}
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {                  // round to nearest even, (lo, hi) packed
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {                   // round to nearest even, (lo, hi) packed
    unsigned r;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
template <bool F16>
__device__ __forceinline__ f16 mma(bf8 a, bf8 b, f16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
// the largest power of two S with bound S < 2^target (bound = m 2^e, 1/2 <= m < 1); exponent clamped so that S and 1 / S are normal
__device__ __forceinline__ float pow2_scale(float bound, int target) {
    const int e = (int)((__float_as_uint(bound) >> 23) & 0xFFu) - 126;
    int se = target - e;
    se = se > 60 ? 60 : se < -60 ? -60 : se;
    return __uint_as_float((unsigned)(127 + se) << 23);
}
__device__ __forceinline__ bf8 xor_bits(bf8 v, unsigned mask) {
    u4 t = __builtin_bit_cast(u4, v);
    t = u4{t.x ^ mask, t.y ^ mask, t.z ^ mask, t.w ^ mask};
    return __builtin_bit_cast(bf8, t);
}

// groups of a unit: (LDS sub-array of the A pieces, of the B pieces).  NPC > 1: piece pairs, b = NPC - 1 .. 0, a = NPC - 1 - b .. 0;
// NPC = 1: the sub-units
template <int NPC>
constexpr int grp_a(int g) { return NPC == 3 ? (g == 0 ? 0 : g == 1 ? 1 : g == 2 ? 0 : 5 - g) : NPC == 2 ? (g == 0 ? 0 : 2 - g) : g; }
template <int NPC>
constexpr int grp_b(int g) { return NPC == 3 ? (g == 0 ? 2 : g < 3 ? 1 : 0) : NPC == 2 ? (g == 0 ? 1 : 0) : g; }
template <int NPC>
constexpr bool grp_last_of_b(int g) { return NPC == 1 ? true : grp_a<NPC>(g) == 0; }

// slot t of a unit applies the sign of an Im piece (four v_xor): such slots get no slice of the build
template <int PWV, int NG>
constexpr bool fix_slot(int t) {
    const int g = t / (6 * PWV), ap = (t / PWV) % 6, cb = t % PWV;
    return cb == PWV - 1 && ((g == 0 && ap < 2) || (g > 0 && ap == 0) || (g + 1 < NG && (ap == 2 || ap == 4)));
}
template <int PWV, int NG>
constexpr int free_slots(int t) {               // slots below t that take slices
    int n = 0;
    for (int i = 0; i < t; ++i) n += fix_slot<PWV, NG>(i) ? 0 : 1;
    return n;
}

}  // namespace gg

// ROWS: where the float32 rows live and how the reverse scan normalised --
//   y_off(tid, c) / yb_off(tid, c): float offset of component c of this thread's (row, clip) inside a step's y / ybar row;
//   rsq(m): 1 / sqrt(m) exactly as the family's reverse kernel computes it.
// Rows of step k: y at stash + ((pair N + k) 2) 4 PD (the y half of the (y, H y) row pair), ybar at gops + (pair N + k) 4 PD.
// LEGACY (round 5, the wide family only): the legacy AudioMPS sums -- Rbar = sum (te_k psi_k) psi_k^dagger + (s ybar) u^dagger with psi_k = u_k
// (the normalised state the step starts from), te_k = 2 (e_k - x_k), s = dt x_k; Qbar = sum ybar u^dagger unchanged.  The first term's B
// operand is u instead of y, its A operand te_k u_k.
template <int PD, int NPC, typename ROWS, bool F16 = false, bool LEGACY = false>
__global__ __launch_bounds__(2 * PD, 1) void k_grad_gemm(Dev P, const float* __restrict__ audio) {
    using namespace gg;
    static_assert(!F16 || NPC == 2, "the fp16 split has two pieces");
    constexpr int PWV = PD / 32;                                  // waves = 32-row blocks
    constexpr int NSUB = NPC == 1 ? 2 : 1;                        // 4-step sub-units per unit
    constexpr int GU = 4 * NSUB;                                  // steps per unit
    constexpr int OPS = 5 * 2 * PD;                               // 16-byte pieces per sub-array: [operand][component][row]
    constexpr int NARR = NPC == 1 ? NSUB : NPC;                   // sub-arrays per buffer
    constexpr int NG = NPC == 1 ? NSUB : NPC * (NPC + 1) / 2;     // groups per unit
    constexpr int NM = NG * 6 * PWV;                              // MFMAs per unit
    // the two operand buffers are two distinct arrays (and the unit loop is unrolled by two): no aliasing between the build's
    // stores and the reads of the unit being multiplied
    __shared__ __attribute__((aligned(16))) u4 opsA[NARR * OPS];
    __shared__ __attribute__((aligned(16))) u4 opsB[NARR * OPS];
    __shared__ __attribute__((aligned(16))) v4 tab[2 * CH * 2];   // [2][CH][2]: (s w, inv, w, te w) per (chunk parity, step, clip)
    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH, NU = (N + GU - 1) / GU;
    const int b0 = 2 * blockIdx.x, b1 = (b0 + 1 < P.B) ? b0 + 1 : b0;
    const bool two = b1 != b0;
    const float A = dev_A(P);
    // ---- build role: this thread = both components of one (row, clip); sixteen lanes = eight rows x two clips (128 contiguous
    //      bytes of an operand array per ds_write_b64) ----
    const int prow = 8 * (tid >> 4) + (tid & 7), pclip = (tid >> 3) & 1;
    const float* stf = reinterpret_cast<const float*>(P.stash) + (size_t)blockIdx.x * N * (8 * PD);   // uniform bases
    const float* ybs = reinterpret_cast<const float*>(P.gops) + (size_t)blockIdx.x * N * (4 * PD);
    float2 ps0 = P.phi0 ? P.phi0[(size_t)((2 * blockIdx.x + pclip) % P.phi_rank) * PD + prow] : P.psi0[prow];   // (RhoCMPS: the column's phi_a)
    // ---- MFMA role: wave w owns the 32-row block w of Re Rbar, Im Rbar, Re Qbar, Im Qbar ----
    const int mr = lane & 31, mh = lane >> 5;
    const unsigned imask = mh ? 0x80008000u : 0u;                 // Im form: K half 1 is -a_re
    const int a_re_off = mh * PD + 32 * w + mr, a_im_off = (mh ^ 1) * PD + 32 * w + mr, b_off = (6 + mh) * PD + mr;

    // fp16 pieces: power-of-two scales of the operand classes (te y | s ybar: sR, ybar: sQ, y | u: sB), see the header
    float sR = 1.f, sQ = 1.f, sB = 1.f;
    if constexpr (F16) {
        float m_s = 0.f, m_t = 0.f, m_n = 1.f;                    // max |s|, max |te| |y|, max |y|^2 over the pair's steps
        for (int e = tid; e < 2 * N; e += 2 * PD) {
            const int idx = e >> 1, cl = e & 1;
            if (cl && !two) continue;
            const float* xr = audio + (size_t)(cl ? b1 : b0) * T;
            const float* sc = P.scal + ((size_t)(cl ? b1 : b0) * NC + idx / CH) * 128;
            const float inc = (idx + 1 < T ? xr[idx + 1] : 0.f) - xr[idx];
            const float nv = sc[idx & (CH - 1)], ev = sc[64 + (idx & (CH - 1))];
            if constexpr (LEGACY) {                               // |psi_k| = 1
                m_s = fmaxf(m_s, fabsf(P.dt * inc));
                m_t = fmaxf(m_t, fabsf(2.0f * (ev - inc)));
            } else {
                const float zbar = -1.0f / (1.0f + (ev * inc) / A);
                m_s = fmaxf(m_s, fabsf(inc / A));
                m_t = fmaxf(m_t, fabsf(2.0f * (zbar * inc / A)) * sqrtf(nv));
            }
            m_n = fmaxf(m_n, nv);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            m_s = fmaxf(m_s, __shfl_xor(m_s, off, 64));
            m_t = fmaxf(m_t, __shfl_xor(m_t, off, 64));
            m_n = fmaxf(m_n, __shfl_xor(m_n, off, 64));
        }
        if (lane == 0) tab[w] = v4{m_s, m_t, m_n, 0.f};
        __syncthreads();
#pragma unroll
        for (int ww = 0; ww < PWV; ++ww) {
            const v4 t = tab[ww];
            m_s = fmaxf(m_s, t.x); m_t = fmaxf(m_t, t.y); m_n = fmaxf(m_n, t.z);
        }
        __syncthreads();
        const float ymax = P.opmax[blockIdx.x];
        auto uni = [](float x) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(x))); };
        sR = uni(pow2_scale(fmaxf(m_t, m_s * ymax), 15));
        sQ = uni(pow2_scale(ymax, 15));
        sB = uni(pow2_scale(sqrtf(m_n), 13));                     // |u| <= 1 <= max |y|
        ps0.x *= sB; ps0.y *= sB;
    }

    f16 Rre[PWV], Rim[PWV], Qre[PWV], Qim[PWV];
#pragma unroll
    for (int cb = 0; cb < PWV; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) Rre[cb][r] = Rim[cb][r] = Qre[cb][r] = Qim[cb][r] = 0.f;

    // per-step scalars of chunk cj (model.py:294, 303 in the reference's operation order, as the reverse scan forms them), threads
    // 0 .. 127 = (step, clip); steps behind the clip's last one (and the repeated clip of an odd batch) get weight 0 and finite
    // scalars, so whatever the build forms from their (clamped) rows multiplies to zero
    auto build_tab = [&](int cj) {
        if (tid < 2 * CH) {
            const int st = tid >> 1, cl = tid & 1, idx = cj * CH + st;
            const bool in = idx < N;
            const float* xr = audio + (size_t)(cl ? b1 : b0) * T;
            const float* sc = P.scal + ((size_t)(cl ? b1 : b0) * NC + cj) * 128;
            const float x0 = idx < T ? xr[idx] : 0.f, x1 = idx + 1 < T ? xr[idx + 1] : 0.f;
            const float inc = x1 - x0;
            const float nv = in ? sc[st] : 1.f, ev = in ? sc[64 + st] : 0.f;
            const float z = (ev * inc) / A;
            const float zbar = -1.0f / (1.0f + z);
            const bool on = in && (cl == 0 || two);
            if constexpr (LEGACY)
                tab[((cj & 1) * CH + st) * 2 + cl] =
                    v4{on ? (P.dt * inc) * sR : 0.f, ROWS::rsq(fmaxf(nv, 1e-12f)) * sB, on ? sQ : 0.f, on ? (2.0f * (ev - inc)) * sR : 0.f};
            else
            tab[((cj & 1) * CH + st) * 2 + cl] =
                v4{on ? (inc / A) * sR : 0.f, ROWS::rsq(fmaxf(nv, 1e-12f)) * sB, on ? sQ : 0.f, on ? (2.0f * (zbar * inc / A)) * sR : 0.f};
        }
    };
    // raw rows of one unit: y_{kb-1 .. kb+GU-1}, ybar_{kb .. kb+GU-1} (both components), rho_{kb-1 .. kb+GU-2}; fetched one unit
    // ahead of their use as buffer loads (descriptor + SGPR row offset + loop-invariant lane offset: no address arithmetic on the
    // VALU).  Row numbers are clamped to the pair's range; row -1 of unit 0 is an in-workspace row whose values are discarded by a
    // select.
    float rY[2][GU + 1], rYB[2][GU];                              // [component][step]
    float2 rRH[GU];
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(stf - 8 * PD), 0, (N + 1) * (8 * PD * 4), 0x00020000);
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ybs), 0, N * (4 * PD * 4), 0x00020000);
    const auto rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(P.rho - PD), 0, (N + 2) * (PD * 8), 0x00020000);
    const int voff_y0 = ROWS::y_off(tid, 0) * 4, voff_y1 = ROWS::y_off(tid, 1) * 4;
    const int voff_b0 = ROWS::yb_off(tid, 0) * 4, voff_b1 = ROWS::yb_off(tid, 1) * 4, voff_r = prow * 8;
    auto load_y = [&](int kb, int c, int j) {
        const int row = (kb - 1 + j) < N ? (kb - 1 + j) : N - 1;
        rY[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, c ? voff_y1 : voff_y0, (row + 1) * (8 * PD * 4), 0));
    };
    auto load_yb = [&](int kb, int c, int j) {
        const int row = (kb + j) < N ? (kb + j) : N - 1;
        rYB[c][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_b, c ? voff_b1 : voff_b0, row * (4 * PD * 4), 0));
    };
    auto load_rho = [&](int kb, int j) {
        const int row = (kb - 1 + j) < N ? (kb - 1 + j) : N;
        rRH[j] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_r, voff_r, (row + 1) * (PD * 8), 0));
    };
    // the loads of sub-unit s of a unit, one by one: the registers its math has just read (the last y row goes with the last sub-unit)
    constexpr int NLD = 22;
    auto load_one = [&](int kb, int s, int l) {
        if (l < 8) load_y(kb, l / 4, 4 * s + l % 4);
        else if (l < 16) load_yb(kb, (l - 8) / 4, 4 * s + (l - 8) % 4);
        else if (l < 20) load_rho(kb, 4 * s + (l - 16));
        else if (s == NSUB - 1) load_y(kb, l - 20, GU);
    };

    // one unit: the MFMAs of unit u from RD (MAC), the operands of unit u + 1 into WR, the raw rows of unit u + 2
    auto run_unit = [&](auto mac_c, const u4* RD, u4* WR, int u) {
        constexpr bool MAC = decltype(mac_c)::value;
        constexpr int NX = NPC == 1 ? 10 : F16 ? 40 : 60;         // split / pack / store slices per sub-unit
        constexpr int BLK = 16 + NLD + NX;                        // slices per sub-unit: math | loads | split, pack, store
        constexpr int NS = 1 + NSUB * BLK;                        // + the table slice
        const int kb = GU * (u + 1);                              // first step of the unit being built
        const int kl = GU * (u + 2);                              // ... of the unit being fetched
        v4 sk[GU];
        float invp0 = 1.f;
        float val[2][GU][5];                                      // te y, s ybar, ybar, y, u per (component, step)
        float t1 = 0.f, t2 = 0.f;
        float sv[2][3];                                           // (x, x - hi, x - hi - mid) of the two steps being packed
        unsigned w0[3] = {0u, 0u, 0u};
        unsigned hp0 = 0u, lp0 = 0u, hp1 = 0u;                   // fp16 pieces of the operand being packed: steps (0, 1) and the hi piece of (2, 3)
        bf8 Areg[6], By[PWV], Bu[PWV];
        auto a_off = [&](int ap) { return (ap < 2 ? 0 : ap < 4 ? 4 * PD : 2 * PD) + ((ap & 1) ? a_im_off : a_re_off); };
        auto read_a = [&](int ap, int arr) { Areg[ap] = __builtin_bit_cast(bf8, RD[(size_t)arr * OPS + a_off(ap)]); };
        auto fix_a = [&](int ap) { Areg[ap] = xor_bits(Areg[ap], imask); };
        auto slice = [&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I == 0) {                               // the unit's table rows (its steps lie in one chunk)
                const v4* tb = tab + (((kb / CH) & 1) * CH + (kb & (CH - 1))) * 2 + pclip;
#pragma unroll
                for (int j = 0; j < GU; ++j) sk[j] = tb[2 * j];
                const int km = kb > 0 ? kb - 1 : 0;
                invp0 = tab[(((km / CH) & 1) * CH + (km & (CH - 1))) * 2 + pclip].y;
            } else {
                constexpr int s = (I - 1) / BLK, r = (I - 1) % BLK;
                if constexpr (r < 16) {                           // math of step j, four parts
                    constexpr int j = 4 * s + r / 4, part = r % 4;
                    if constexpr (part == 0) {                    // u = rho_{k-1} y_{k-1} / |y_{k-1}| (psi0 at k = 0)
                        const float invp = j == 0 ? invp0 : sk[j > 0 ? j - 1 : 0].y;
                        t1 = rY[0][j] * invp;
                        t2 = rY[1][j] * invp;
                        const float ur = rRH[j].x * t1 - rRH[j].y * t2;
                        val[0][j][4] = (kb + j > 0) ? ur : ps0.x;
                    } else if constexpr (part == 1) {
                        const float ui = rRH[j].x * t2 + rRH[j].y * t1;
                        val[1][j][4] = (kb + j > 0) ? ui : ps0.y;
                    } else {
                        constexpr int c = part - 2;
                        val[c][j][0] = LEGACY ? sk[j].w * (F16 ? val[c][j][4] * (1.0f / sB) : val[c][j][4]) : sk[j].w * rY[c][j + 1];
                        val[c][j][1] = sk[j].x * rYB[c][j];
                        val[c][j][2] = sk[j].z * rYB[c][j];
                        val[c][j][3] = F16 ? rY[c][j + 1] * sB : rY[c][j + 1];
                    }
                } else if constexpr (r < 16 + NLD) {              // one load of the unit after
#if !(defined(CMPS_DIAG) && defined(WABL_GRAD_NO_LOADS))          // diagnostic builds only (results are wrong)
                    load_one(kl, s, r - 16);
#endif
                } else if constexpr (NPC == 1) {                  // round one operand's four steps to bf16, store
                    constexpr int x = r - 16 - NLD, c = x / 5, o = x % 5;
                    unsigned* d = reinterpret_cast<unsigned*>(WR + (size_t)s * OPS + (o * 2 + c) * PD + prow) + 2 * pclip;
                    *reinterpret_cast<uint2*>(d) = make_uint2(cvt_pk_bf16(val[c][4 * s][o], val[c][4 * s + 1][o]),
                                                              cvt_pk_bf16(val[c][4 * s + 2][o], val[c][4 * s + 3][o]));
                } else {                                          // split two steps of one operand, pack, store
                    constexpr int x = r - 16 - NLD, c = x / 30, o = (x / 6) % 5, jp = (x / 3) % 2, part = x % 3;
                    if constexpr (F16) {
                        // two slices per pair of steps: hi piece + residuals | lo piece (and, behind the second pair, the two 8-byte stores)
                        constexpr int xf = r - 16 - NLD, cf = xf / 20, of = (xf / 4) % 5, jpf = (xf / 2) % 2, partf = xf % 2;
                        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                        if constexpr (partf == 0) {
                            const float v0 = val[cf][4 * s + 2 * jpf][of], v1 = val[cf][4 * s + 2 * jpf + 1][of];
                            const unsigned hp = cvt_pk_f16(v0, v1);
                            const h2 hh = __builtin_bit_cast(h2, hp);
                            if constexpr (jpf == 0) hp0 = hp; else hp1 = hp;
                            sv[0][1] = v0 - (float)hh.x;
                            sv[1][1] = v1 - (float)hh.y;
                        } else if constexpr (jpf == 0) {
                            lp0 = cvt_pk_f16(sv[0][1], sv[1][1]);
                        } else {
                            const unsigned lp1 = cvt_pk_f16(sv[0][1], sv[1][1]);
                            unsigned* d0 = reinterpret_cast<unsigned*>(WR + (of * 2 + cf) * PD + prow) + 2 * pclip;
                            unsigned* d1 = reinterpret_cast<unsigned*>(WR + (size_t)OPS + (of * 2 + cf) * PD + prow) + 2 * pclip;
                            *reinterpret_cast<uint2*>(d0) = make_uint2(hp0, hp1);
                            *reinterpret_cast<uint2*>(d1) = make_uint2(lp0, lp1);
                        }
                    } else if constexpr (part == 0) {             // the two steps' chains interleaved: back-to-back dependent VALU
                        const float v0 = val[c][4 * s + 2 * jp][o], v1 = val[c][4 * s + 2 * jp + 1][o];   // do not hide behind an MFMA
                        const unsigned h0 = __float_as_uint(v0) & 0xFFFF0000u, h1 = __float_as_uint(v1) & 0xFFFF0000u;
                        sv[0][0] = v0; sv[1][0] = v1;
                        sv[0][1] = v0 - __uint_as_float(h0);
                        sv[1][1] = v1 - __uint_as_float(h1);
                    } else if constexpr (part == 1) {
                        const unsigned m0 = __float_as_uint(sv[0][1]) & 0xFFFF0000u, m1 = __float_as_uint(sv[1][1]) & 0xFFFF0000u;
                        sv[0][2] = sv[0][1] - __uint_as_float(m0);
                        sv[1][2] = sv[1][1] - __uint_as_float(m1);
                    } else {
                        unsigned pk[3];
#pragma unroll
                        for (int a = 0; a < 3; ++a) pk[a] = pack_hi16(__float_as_uint(sv[0][a]), __float_as_uint(sv[1][a]));
                        if constexpr (jp == 0) {
#pragma unroll
                            for (int a = 0; a < 3; ++a) w0[a] = pk[a];
                        } else {
#pragma unroll
                            for (int a = 0; a < NPC; ++a) {
                                unsigned* d = reinterpret_cast<unsigned*>(WR + (size_t)a * OPS + (o * 2 + c) * PD + prow) + 2 * pclip;
#if defined(CMPS_DIAG) && defined(WABL_GRAD_NO_STORE)
                                if constexpr (MAC) { asm volatile("" : : "v"(w0[a]), "v"(pk[a])); continue; }
#endif
                                *reinterpret_cast<uint2*>(d) = make_uint2(w0[a], pk[a]);
                            }
                        }
                    }
                }
            }
        };
        if constexpr (MAC) {
            // operands of the first group
            constexpr int a0 = grp_a<NPC>(0), bb0 = grp_b<NPC>(0);
            read_a(0, a0);
#pragma unroll
            for (int cb = 0; cb < PWV; ++cb) By[cb] = __builtin_bit_cast(bf8, RD[(size_t)bb0 * OPS + b_off + 32 * cb]);
            read_a(1, a0);
            read_a(2, a0);
#pragma unroll
            for (int cb = 0; cb < PWV; ++cb) Bu[cb] = __builtin_bit_cast(bf8, RD[(size_t)bb0 * OPS + b_off + 2 * PD + 32 * cb]);
            read_a(3, a0);
            read_a(4, a0);
            read_a(5, a0);
            fix_a(1);
            __builtin_amdgcn_sched_barrier(0);
        }
        static_for<0, (MAC ? NM : NS)>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (MAC) {
                constexpr int g = t / (6 * PWV), ap = (t / PWV) % 6, cb = t % PWV;
                constexpr int ng = g + 1;
                if constexpr (ap == 0) Rre[cb] = mma<F16>(Areg[0], LEGACY ? Bu[cb] : By[cb], Rre[cb]);
                if constexpr (ap == 1) Rim[cb] = mma<F16>(Areg[1], LEGACY ? Bu[cb] : By[cb], Rim[cb]);
                if constexpr (ap == 2) Qre[cb] = mma<F16>(Areg[2], Bu[cb], Qre[cb]);
                if constexpr (ap == 3) Qim[cb] = mma<F16>(Areg[3], Bu[cb], Qim[cb]);
                if constexpr (ap == 4) Rre[cb] = mma<F16>(Areg[4], Bu[cb], Rre[cb]);
                if constexpr (ap == 5) Rim[cb] = mma<F16>(Areg[5], Bu[cb], Rim[cb]);
                if constexpr (cb == PWV - 1) {
                    // the Im pieces carry the sign of their K half: applied a row of MFMAs after the read was issued
                    if constexpr (g == 0 && ap == 0) fix_a(3);                                 // (first group: read before slot 0)
                    if constexpr (g == 0 && ap == 1) fix_a(5);
                    if constexpr (g > 0 && ap == 0) fix_a(5);                                  // refilled at the end of the group before
                    if constexpr (ng < NG) {
                        read_a(ap, grp_a<NPC>(ng));                                            // this A register is free: next group's piece
                        if constexpr (ap == 2) fix_a(1);
                        if constexpr (ap == 4) fix_a(3);
                    }
                }
                if constexpr (ng < NG && grp_last_of_b<NPC>(g)) {                              // last group of these B pieces: next ones
                    constexpr int nb = grp_b<NPC>(ng < NG ? ng : 0);
                    if constexpr (ap == 1) By[cb] = __builtin_bit_cast(bf8, RD[(size_t)nb * OPS + b_off + 32 * cb]);
                    if constexpr (ap == 5) Bu[cb] = __builtin_bit_cast(bf8, RD[(size_t)nb * OPS + b_off + 2 * PD + 32 * cb]);
                }
                __builtin_amdgcn_sched_barrier(0);
#if !(defined(CMPS_DIAG) && defined(WABL_GRAD_NO_SLICES))          // diagnostic builds only (results are wrong): the MFMA stream alone
                constexpr int NF = free_slots<PWV, NG>(NM), f0 = free_slots<PWV, NG>(t);
                if constexpr (!fix_slot<PWV, NG>(t)) static_for<(f0 * NS) / NF, ((f0 + 1) * NS) / NF>(slice);
#endif
            } else {
                slice(tc);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    // chunk tables: chunk c + 1 is built at the second unit of chunk c (the chunk below c is no longer read by then) and is
    // first read several units (barriers) later
    build_tab(0);
#pragma unroll
    for (int s = 0; s < NSUB; ++s)
#pragma unroll
        for (int l = 0; l < NLD; ++l) load_one(0, s, l);
    __syncthreads();
    run_unit(std::false_type{}, opsB, opsA, -1);                  // operands of unit 0, raw rows of unit 1
    __syncthreads();
    for (int u = 0; u < NU; u += 2) {                             // an odd count runs one unit of zero operands
        run_unit(std::true_type{}, opsA, opsB, u);
        __syncthreads();
        run_unit(std::true_type{}, opsB, opsA, u + 1);
        if (((u + 1) & (CH / GU - 1)) == 1) build_tab((u + 1) / (CH / GU) + 1);
        __syncthreads();
    }

    float* slab = P.slabs + (size_t)blockIdx.x * P.slab_floats;
    constexpr int DD = PD * PD;
    const float iR = 1.0f / sR, iQ = 1.0f / sQ, iB = 1.0f / sB;     // exact (powers of two); 1 without the fp16 scales
#pragma unroll
    for (int cb = 0; cb < PWV; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * mh;    // C/D layout of the 32x32 MFMA: column = lane & 31
            const int o = row * PD + 32 * cb + mr;
            slab[o] = F16 ? Rre[cb][r] * iR * iB : Rre[cb][r];
            slab[DD + o] = F16 ? Rim[cb][r] * iR * iB : Rim[cb][r];
            slab[2 * DD + o] = F16 ? Qre[cb][r] * iQ * iB : Qre[cb][r];
            slab[3 * DD + o] = F16 ? Qim[cb][r] * iQ * iB : Qim[cb][r];
        }
}

}  // namespace cmps
