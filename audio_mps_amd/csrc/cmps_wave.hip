// Wave-per-clip kernels (the hot path for D <= 32): one 64-lane wavefront owns one clip for the whole
// scan.  This file holds the reverse scan and the sampler; the forward scan (two waves per clip) is cmps_wave2.hip.  Measured on MI355X (scripts/ubench/issue_rate.hip): a lone wave issues one VALU instruction per
// ~5.3 cycles whether it is v_fma_f32 or v_pk_fma_f32, dependent accumulation chains cost nothing extra,
// and the fp32 MFMA shares the fp32 ALUs with the VALU (no overlap).  The kernels are therefore
// instruction-issue bound; the design minimises instruction count and keeps every memory latency off the
// instruction stream.
//
// Lane layout (DP = 32, smaller D zero-padded): lane l = (i = l & 31, h = l >> 5).
//   * SPLIT layout of a complex D-vector X: ONE float per lane, x = h ? Im X_i : Re X_i.
//   * where a lane needs the whole complex number it holds the pair (own, osig) with
//       osig = h ? -Re X_i : Im X_i,   i.e. the pair is X_i in half 0 and -i X_i in half 1.
//     Multiplying such pairs by an ordinary complex scalar (the rotation rho) is plain complex pair
//     arithmetic in every lane; products between two lane-held vectors use the split form:
//     sum over all 64 lanes of a_own * b_own = Re(a^dagger b).
//   * lane (i, h) holds columns 16h..16h+15 of row i of each matrix as 16 (re, im) pairs (32 VGPRs).
//     mat-vec: the vector is broadcast through LDS (4-byte write per lane, 8 x ds_read_b128 per half),
//     32 v_pk_fma_f32 per lane -- a complex multiply-accumulate is two packed FMAs that use op_sel /
//     neg_lo on the SAME register pair -- then ONE v_permlane32_swap + ONE add combines the two halves
//     and lands the result directly in the split layout.
//   * wave reductions: 4 fused DPP adds inside 16-lane rows + 2 row_bcast adds + v_readlane.
//   * per-step tables (rotation rho_k, and in the reverse sweep the stashed y_k and H y_k) are staged in LDS
//     one chunk (64 / 32 steps) at a time; the next chunk is prefetched into registers while the current one
//     is consumed, so the inner loops contain no global loads and wait on no memory latency.
//   * the LDS round trip of a broadcast is ~350 cycles when four waves share the LDS: the forward keeps only
//     ONE of them on its serial chain (u -> y), starts the FMAs on the first half of the reads, and runs the
//     mat-vec that only feeds the loss (H y, e = y^dagger H y) one step late, reading y back from LDS.
//   * everything uniform across lanes and off the serial chain (x/A, (e x)/A, log(1+z), 1/(1+z), ...) is
//     evaluated once per 64 steps, one step per lane, in the reference's operation order, and fetched per
//     step with v_readlane.  The float32 loss accumulation stays strictly sequential in time (model.py:279).
//   * backward: R y and R^dagger y only enter as their sum H y (H = R + R^dagger), which the forward already
//     computes for e = y^dagger H y and stashes, so the reverse step has TWO mat-vecs (Q ybar, R^dagger ybar);
//     the three rank-1 gradient updates per step are six v_mfma_f32_32x32x2_f32 (exact fp32; K = 2 is
//     {real, imaginary}; the split layout IS the MFMA operand layout); the part of the adjoint that does not
//     depend on the incoming cotangent is computed one step ahead, off the serial chain.
// Recurrence and adjoint: see the header of cmps_block.hip (same arithmetic, same reference lines).
#include "cmps_wave_util.h"

namespace cmps {

// ------------------------------------------------------------------------------------------------
// backward
// ------------------------------------------------------------------------------------------------
namespace {

// Everything about step j that does not depend on the incoming cotangent (split layout, see header).
struct Pre {
    float yh;      // yhat_j = y_j / sqrt(max(n_j, 1e-12))           (split)
    float yho;     // its osig
    float yhp;     // yhat_j, or 0 when n_j <= 1e-12 (the projection term vanishes)
    float un;      // u_{j+1} = rho_j yhat_j                          (split)
    float uno;     // its osig
    float pre;     // 2 ebar_j ((R + R^dagger) y_j)                   (split)
    v2f rho;       // rho_j
    float inv, s, ten, dtk;   // 1/sqrt(max(n_j,1e-12)), x_j/A, 2 ebar_j n_j, t_j - t_{j+1}
    float rad;     // 2 ebar_j e_j = Re(u_j^dagger g_j), the radial derivative at u_j (Euler: e is homogeneous of degree 2)
};

}  // namespace

// RANK1: how the three rank-1 gradient updates of a step are applied (cmps_set_option(CMPS_OPT_RANK1)):
//   0 exact fp32 MFMA every step; 1 bf16 hi/lo split, 3 products; 2 bf16 hi/mid/lo split, 6 products; 3 (round 4) two fp16 pieces,
//   round to nearest (11 + 1 + 11 + 1 bits: BF16X3's accuracy class), 3 products on v_mfma_f32_32x32x16_f16, with power-of-two
//   scales per eight-step octet from a guaranteed bound of |ybar| (see below)
// LEGACY: the reverse sweep of the previous-generation AudioMPS arithmetic (cmps_legacy.hip) on the same chain.  With rho = 1 in the
// tables, g = cotangent of psi_{k+1} without its own loss term, and e_k = psi_k^dagger H psi_k (degree 0 in y_{k-1}):
//   ybar_k = (g_{k+1} - yhat_k dot) inv_k + te_{k+1} inv_k^2 (H y_k),   dot = Re(yhat_k^dagger g_{k+1}) + te_{k+1} e_{k+1}
//   (Re(yhat_k^dagger g_{k+1}) is 0 analytically -- everything downstream of psi_{k+1} but e_{k+1} is scale invariant -- and is
//   taken exactly once per staged chunk);   g_k = ybar_k + (Q^dagger + dt x_k R^dagger) ybar_k;   te_k = 2 (e_k - x_k)
//   Rbar += (te_k psi_k) psi_k^dagger + (dt x_k ybar_k) psi_k^dagger,   Qbar += ybar_k psi_k^dagger
// so only the per-step scalar rows, Q^dagger instead of Q and the first rank-1 operand differ.
template <int RANK1, bool LEGACY = false>
__global__ __launch_bounds__(64 * WAVES, 1) void k_bwd_wave(Dev P, const float* __restrict__ audio) {
    __shared__ __attribute__((aligned(16))) float4 stY[WAVES][CHB * 32];   // stashed (y, H y) rows of the staged chunk
    __shared__ __attribute__((aligned(16))) float4 stR[WAVES][CHB * 16];   // rho rows
    __shared__ __attribute__((aligned(16))) float4 scl[WAVES][CH * 2];     // per-step scalars, one 32-B row per step
    __shared__ __attribute__((aligned(16))) float2 bcB[WAVES][DPW];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    const int b = blockIdx.x * WAVES + w;      // wave-uniform
    if (b >= P.B) return;
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH;
    stagger(w);

    // One mat-vec on the chain: g = ybar + Q ybar + s_k R^dagger ybar = ybar + M_k ybar, M_k = Q + s_k R^dagger formed with
    // one packed FMA per complex entry in the shadow of the step's LDS broadcast (see cmps_wave2.hip).
    v2f MRd[16], MQ[16], MM[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const v2f rt = ld2(&P.RT[i * DPW + 16 * h + m]);   // R[16h+m][i]
        MRd[m] = mk2(rt.x, -rt.y);                          // R^dagger[i][16h+m]
        if constexpr (LEGACY) {
            const v2f qt = ld2(&P.QT[i * DPW + 16 * h + m]);
            MQ[m] = mk2(qt.x, -qt.y);                       // Q^dagger (the legacy Q is not Hermitian)
        } else {
            MQ[m] = ld2(&P.Q[i * DPW + 16 * h + m]);
        }
    }
    v16f Rre = {}, Rim = {}, Qre = {}, Qim = {};
    // RANK1 == 3: power-of-two scales of the split operands.  a1 | a2 = s ybar share sR (both land in Rre / Rim), ybar has sQ, the
    // unit vectors yhat, u are scaled by 2^13.  sR, sQ follow a GUARANTEED bound of the octet about to run: with the affine maps
    //   f_k(x) = a_k x + c_k,   a_k = inv_k (1 + |Q|_F + |s_{k+1}| |R|_F),   c_k = |te_k| 2 |R|_F |y_k| + |rad_{k+1}| invok_k |y_k| inv_k
    // (2-norms: |ybar_k| <= a_k |ybar_{k+1}| + c_k) the eight steps of an octet satisfy |ybar_k| <= A_k Y + C_k, (A_k, C_k) = f_k o ... o
    // f_top of the octet (a segmented suffix scan over the lanes of scal_commit) and Y = |ybar| of the step above the octet, MEASURED.
    // (The horizon has to be this short: with the reference's initialisation |s| |R|_F is of order one, and a bound compounded over a
    // 64-step chunk is loose by tens of binades -- the first version flushed everything to zero on tests/golden/reftest_d7_t256_b8.)
    // The scales change only when the bound leaves [2^8, 2^15) in scaled units; the accumulators are then multiplied by the exact ratio.
    constexpr float SB16 = 8192.f;
    float sR = 1.f, sQ = 1.f;                     // current scales (wave-uniform)
    float cbA = 0.f, cbC = 0.f, cbsA = 0.f, cbsC = 0.f, cbT = 0.f;   // per octet of the chunk last committed (lane 8 o + .. holds octet o): max A, C, |s| A, |s| C, |ten|
    float s_above = 0.f, rad_above = 0.f;         // s, rad of the first step of the chunk above (no step N: 0)
    float a_above = 1.f, c_above = 0.f, as_above = 0.f, t_above = 0.f;   // the bound's (a, c, |s|, |ten|) of that step (nothing above the top chunk: identity)
    float ybar_last = 0.f;                        // ybar of the step the chain did last (this lane's component)
    float Qn = 0.f, Rn = 0.f;
    if constexpr (RANK1 == 3) {
        float q2 = 0.f, r2 = 0.f;
#pragma unroll
        for (int m = 0; m < 16; ++m) { q2 += MQ[m].x * MQ[m].x + MQ[m].y * MQ[m].y; r2 += MRd[m].x * MRd[m].x + MRd[m].y * MRd[m].y; }
        Qn = 1.001f * sqrtf(sum64(q2));
        Rn = 1.001f * sqrtf(sum64(r2));
    }

    const unsigned aBw = lds_addr(&bcB[w][0]) + i * 8 + h * 4, aBr = lds_addr(&bcB[w][0]) + h * 128;
    const unsigned aYown = lds_addr(&stY[w][0]) + (2 * i + h) * 8;   // stash rows: 64 (y[n], (H y)[n]) pairs, n = 2 i + {re, im}
    const unsigned aRho = lds_addr(&stR[w][0]) + i * 8;
    const unsigned aScl = lds_addr(&scl[w][0]);
    const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
    // RhoCMPS on virtual clips (round 5; Dev::phi0): wave b is column av of clip bc -- the clip's audio and per-step scalars (tr rho', e), the
    // column's rows in the [clip][step][column] stash of k_fwd_rho_mfma (the same 512-B row format), phi_av as the initial vector
    const int vr = P.phi0 ? P.phi_rank : 1;
    const int bc = b / vr, av = b - bc * vr;
    const float4* sty4 = reinterpret_cast<const float4*>(P.hst + ((size_t)bc * N * vr + av) * 128);
    const float* xrow = audio + (size_t)bc * T;
    const float* sc = P.scal + scal_off(bc, NC, 0);
    const float A = dev_A(P);

    float facc = 0.f;   // per lane: sum_k dtk * g_osig * u_own   (the two halves are added at the end)
    float accS = 0.f;   // per lane: sum_k (M_k ybar_k)_own u_own = sum_k Re(u^dagger Q ybar) + s_k Re(u^dagger R^dagger ybar); the Q part
                        // equals Re tr(Q Qbar^T...) and is removed by k_finalize from the reduced Qbar (Dev::abar_fix)
    float accA = 0.f;   // per lane (one step per lane): sum_k zbar_k (e_k x_k)

    // per-step scalars of the 64-step chunk the "pre" stage is working in: computed one step per lane, then
    // written to LDS as one 32-B row per step (s, dtk, inv, ten | te, invok, rad, -) so that a step fetches all
    // of them with two broadcast reads instead of seven v_readlane
    float ra0 = 0.f, ra1 = 0.f, rdt = 0.f, rnv = 1.f, rev = 0.f;   // raw prefetched values of the next chunk
    v4f sry[16], srr[8];
    auto scal_load = [&](int c) {
        const int idx = c * CH + lane;
        ra0 = idx < T ? xrow[idx] : 0.f;
        ra1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
        rdt = P.dtk[idx];                 // padded to N + 64 entries
        rnv = sc[(size_t)c * 128 + lane];
        rev = sc[(size_t)c * 128 + 64 + lane];
    };
    float te_above = 0.f;             // LEGACY: te of the first step of the chunk above (no step N: 0)
    auto scal_commit = [&](int c) {
        const int idx = c * CH + lane;
        const float inc = ra1 - ra0;
        const float nv = rnv, ev = rev;
        const float invv = rsq_nr(fmaxf(nv, 1e-12f));
        const float invokv = nv > 1e-12f ? invv : 0.f;
        if constexpr (LEGACY) {
            const float tev = idx < N ? 2.0f * (ev - inc) : 0.f;              // te_k = 2 ebar_k, ebar_k = e_k - x_k
            float ten = __shfl_down(tev, 1, 64);                              // te_{k+1}
            if (lane == 63) ten = te_above;
            te_above = rdlane(tev, 0);
            // row: (s, dtk, inv, te_k | te_{k+1} inv_k^2 (the coefficient of H y_k), invok, rad = te_k e_k, -)
            scl[w][2 * lane] = make_float4(P.dt * inc, rdt, invv, tev);
            scl[w][2 * lane + 1] = make_float4(ten * invv * invv, invokv, tev * ev, 0.f);
            return;
        }
        const float sv = inc / A;
        const float ex = ev * inc;                      // model.py:294 operation order
        const float z = ex / A;
        const float zbar = -1.0f / (1.0f + z);
        const float ebar = zbar * inc / A;
        const float tev = 2.0f * ebar;
        scl[w][2 * lane] = make_float4(sv, rdt, invv, tev * nv);
        scl[w][2 * lane + 1] = make_float4(tev, invokv, tev * ev, 0.f);
        if (idx < N) accA += zbar * ex;
        if constexpr (RANK1 == 3) {
            const bool in = idx < N;
            const float radv = in ? tev * ev : 0.f;              // (rows behind the clip's last step hold whatever the workspace held: no step N, rad_N = 0)
            float s_up = __shfl_down(sv, 1, 64), rad_up = __shfl_down(radv, 1, 64);     // step j + 1
            if (lane == 63) { s_up = s_above; rad_up = rad_above; }
            s_above = rdlane(sv, 0); rad_above = rdlane(radv, 0);
            const float ynorm = 1.001f * sqrtf(fmaxf(nv, 1e-12f));
            const float aj = in ? invv * (1.0f + Qn + fabsf(s_up) * Rn) : 1.f;
            const float cj = in ? fabsf(tev) * (2.0f * Rn) * ynorm + fabsf(rad_up) * invokv * (ynorm * invv) : 0.f;
            const float sj = in ? fabsf(sv) : 0.f, tj = in ? fabsf(tev * nv) : 0.f;
            // The loop below runs the CHAIN one step ahead of the index its chunks are counted in (chain step k = j + 1): between two
            // boundaries the chain does steps 64 c + 64 down to 64 c + 1.  The set of a chunk is therefore its own steps but the lowest,
            // plus the lowest step of the chunk above (carried from that chunk's commit); position p <-> step 64 c + 1 + p.
            float Aj = __shfl_down(aj, 1, 64), Cj = __shfl_down(cj, 1, 64), as = __shfl_down(sj, 1, 64), m4 = __shfl_down(tj, 1, 64);
            if (lane == 63) { Aj = a_above; Cj = c_above; as = as_above; m4 = t_above; }
            a_above = rdlane(aj, 0); c_above = rdlane(cj, 0); as_above = rdlane(sj, 0); t_above = rdlane(tj, 0);
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {             // suffix scan inside each octet, from its top position down: f_p o (f_{p+1} o ...)
                const float Ag = __shfl_down(Aj, d, 64), Cg = __shfl_down(Cj, d, 64);
                if ((lane & 7) + d < 8) { Cj = fmaf(Aj, Cg, Cj); Aj *= Ag; }
            }
            float m0 = Aj, m1 = Cj, m2 = as * Aj, m3 = as * Cj;
#pragma unroll
            for (int off = 4; off > 0; off >>= 1) {       // maxima over the octet's eight positions, in every lane of the octet
                m0 = fmaxf(m0, __shfl_xor(m0, off, 64)); m1 = fmaxf(m1, __shfl_xor(m1, off, 64)); m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
                m3 = fmaxf(m3, __shfl_xor(m3, off, 64)); m4 = fmaxf(m4, __shfl_xor(m4, off, 64));
            }
            cbA = m0; cbC = m1; cbsA = m2; cbsC = m3; cbT = m4;
        }
    };
    // RANK1 == 3: scales of the chunk just committed from |ybar| of the step above it; the accumulators follow (exact ratios)
    const int f16_shift = P.f16_shift;                 // CMPS_OPT_F16_SCALE_SHIFT: 0 but in the test that provokes CMPS_ERR_F16_RANGE
    auto pow2_of = [f16_shift](float bound) {          // the largest power of two S with bound S < 2^15 (exponent clamped)
        int se = 15 + f16_shift - ((int)((__float_as_uint(bound) >> 23) & 0xFFu) - 126);
        se = se > 60 ? 60 : se < -60 ? -60 : se;
        return __uint_as_float((unsigned)(127 + se) << 23);
    };
    auto pow2_inv = [](float sc2) { return __uint_as_float(0x7F000000u - __float_as_uint(sc2)); };
    // before an octet (chain steps 8 m + 8 .. 8 m + 1; `lane8` = any lane of the octet's group in the committed chunk): bounds from |ybar|
    // of the step above, and -- only if a bound has left the window -- new scales, the pending MFMAs issued under the old ones, the
    // accumulators multiplied by the exact ratios.  Returns with the fragments cleared when it rescaled (the hooks are unconditional).
    auto scale_window = [](float bound, float sc) {      // true if bound * sc is outside [2^8, 2^15) (or sc was never set)
        const float x = bound * sc;
        return !(x >= 256.f && x < 32768.f);
    };
    float nRs = 1.f, nQs = 1.f;
    auto octet_bounds = [&](int lane8, float Y) -> bool {  // true: the scales have to change (new ones in nRs, nQs)
        const float bQ = fmaf(rdlane(cbA, lane8), Y, rdlane(cbC, lane8));
        const float bR = fmaxf(rdlane(cbT, lane8), fmaf(rdlane(cbsA, lane8), Y, rdlane(cbsC, lane8)));
        const bool chQ = bQ > 0.f && scale_window(bQ, sQ), chR = bR > 0.f && scale_window(bR, sR);
        nQs = chQ ? pow2_of(bQ * 8.f) : sQ;               // the bound lands in [2^11, 2^12): room both ways
        nRs = chR ? pow2_of(bR * 8.f) : sR;
        return chQ || chR;
    };
    auto apply_scales = [&]() {
        const float fR = nRs * pow2_inv(sR), fQ = nQs * pow2_inv(sQ);
#pragma unroll
        for (int r = 0; r < 16; ++r) { Rre[r] *= fR; Rim[r] *= fR; Qre[r] *= fQ; Qim[r] *= fQ; }
        sR = nRs; sQ = nQs;
    };
    auto stage_load_all = [&](int hh) {
        stage_load512<16>(sty4, hh * CHB, N - 1, lane, sry, vr);
        stage_load<8>(rho4, hh * CHB, N, lane, srr);
    };
    auto stage_commit_all = [&]() {
        stage_commit<16>(stY[w], lane, sry);
        stage_commit<8>(stR[w], lane, srr);
    };

    // the off-chain stage for step j; its LDS reads were issued earlier by own_issue
    auto make_pre = [&](v2f yh2, v2f rho, v4f c0, v4f c1) -> Pre {
        const float yown = yh2.x, hown = yh2.y;
        Pre S;
        S.rho = rho;
        S.s = c0.x;
        S.dtk = c0.y;
        S.inv = c0.z;
        S.ten = c0.w;
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

    const int hl = (N - 1) / CHB;
    stage_load_all(hl);
    scal_load(hl >> 1);
    stage_commit_all();
    scal_commit(hl >> 1);
    v4f qc[8];
    v2f yh_j, rho_j;
    v4f c0_j, c1_j;
    Pre S;
    {
        const int jr = (N - 1) & (CHB - 1), jc = (N - 1) & (CH - 1);
        own_issue(aYown + jr * 512, aRho + jr * 256, aScl + jc * 32, yh_j, rho_j, c0_j, c1_j);
        lds_wait_own<0>(yh_j, rho_j, c0_j, c1_j);
        S = make_pre(yh_j, rho_j, c0_j, c1_j);
    }
    float g = 0.f, go = 0.f;                      // cotangent of u_{k+1}: split value and its osig
    const float2 p0 = P.phi0 ? P.phi0[av * DPW + i] : P.psi0[i];
    const float u0 = hb ? p0.y : p0.x, u0o = hb ? -p0.x : p0.y;

    // one step of the serial chain (step k = the step S describes); with have_pre, the pre stage of step
    // k-1 (index jn) runs in its shadow and supplies u_k.
    // The normalisation adjoint needs dot = Re(yhat_k^dagger conj(rho_k) g) = Re(u_{k+1}^dagger g_{k+1}), the radial
    // derivative at u_{k+1}.  Everything downstream of u_{k+1} is scale invariant except the loss term of step
    // k+1, which is homogeneous of degree 2 in u_{k+1}, so dot = 2 e_{k+1} ebar_{k+1} exactly (rad_next).  Using
    // it keeps the 144-cycle wave reduction off the serial chain; `exact` (once per staged chunk) does the real
    // projection so that rounding in the radial direction cannot accumulate over the clip.
    float rad_next = 0.f;                         // no step N: g_N = 0
    // Rank-1 updates, three forms (template parameter RANK1, chosen at run time by cmps_set_option):
    //   0  EXACT_F32: six exact-fp32 v_mfma_f32_32x32x2_f32 per step (the fp32 MFMA shares the fp32 ALUs with the
    //      VALU, so each holds the wave for 64 cycles).
    //   1  BF16X2: the seven operand values of a step are split into bf16 hi (truncation) + bf16 lo (rounded
    //      remainder) and packed into slot P of per-lane K-fragments; after eight steps 18 v_mfma_f32_32x32x16_bf16
    //      (hi*hi + hi*lo + lo*hi per product; K = 8 steps x {re, im}) accumulate them into the same fp32 tiles:
    //      16 operand bits, error <= ~2^-16 |a||b| per product.
    //   2  BF16X3: exact three-way split x = hi + mid + lo (8 + 8 + 8 significand bits, all by truncation, so the
    //      decomposition is exact) and SIX products per pair (hi*hi, hi*mid, mid*hi, hi*lo, lo*hi, mid*mid): 24 operand
    //      bits; the dropped terms (mid*lo, lo*mid, lo*lo) are <= 2^-23 |a||b|, i.e. the product is fp32-faithful.  36 MFMAs
    //      per octet, applied during the next two steps.
    // The bf16 MFMA runs on the matrix pipe beside the VALU; accumulation is fp32 in all three forms.
    unsigned fH[7][4], fL[7][4];      // K-fragments: value v, register r holds slots 2r (low half) and 2r+1 (high half)
    unsigned fM[RANK1 == 2 ? 7 : 1][4];
    float fsave[7], hold_e[7], hold_o[7];
    auto split_pair = [&](int v, int reg, float ve, float vo) {
        if constexpr (RANK1 == 3) {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const float sc2 = (v == 0 || v == 2) ? sR : v == 1 ? sQ : SB16;
            const float te_ = ve * sc2, to_ = vo * sc2;
            unsigned hi, lo;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(te_), "v"(to_));
            const h2 hh = __builtin_bit_cast(h2, hi);
            const float re_ = te_ - (float)hh.x, ro_ = to_ - (float)hh.y;
            asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo) : "v"(re_), "v"(ro_));
            fH[v][reg] = hi;
            fL[v][reg] = lo;
            return;
        }
        const unsigned xe = __float_as_uint(ve), xo = __float_as_uint(vo);
        fH[v][reg] = __builtin_amdgcn_perm(xo, xe, 0x07060302u);
        v2f lo;
        lo.x = ve - __uint_as_float(xe & 0xFFFF0000u);
        lo.y = vo - __uint_as_float(xo & 0xFFFF0000u);
        if constexpr (RANK1 == 2) {
            const unsigned me = __float_as_uint(lo.x), mo = __float_as_uint(lo.y);
            fM[v][reg] = __builtin_amdgcn_perm(mo, me, 0x07060302u);
            v2f l3;
            l3.x = lo.x - __uint_as_float(me & 0xFFFF0000u);
            l3.y = lo.y - __uint_as_float(mo & 0xFFFF0000u);
            // at most 8 significand bits are left: the truncation to bf16 is exact
            fL[v][reg] = __builtin_amdgcn_perm(__float_as_uint(l3.y), __float_as_uint(l3.x), 0x07060302u);
        } else {
            fL[v][reg] = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, bf2));
        }
    };
    // Slots 7 and 6 of an octet are held raw and split only at slot 4: fragment register 3 of the PREVIOUS octet then stays
    // intact through the steps of slots 7, 6 and 5, the window in which that octet's MFMAs are issued (below).
    auto record = [&](auto slot, const float (&val)[7]) {
        constexpr int PSLOT = decltype(slot)::value;
        if constexpr (PSLOT & 1) {
#pragma unroll
            for (int v = 0; v < 7; ++v) fsave[v] = val[v];
        } else if constexpr (PSLOT == 6) {
#pragma unroll
            for (int v = 0; v < 7; ++v) { hold_e[v] = val[v]; hold_o[v] = fsave[v]; }
        } else {
            if constexpr (PSLOT == 4) {
#pragma unroll
                for (int v = 0; v < 7; ++v) split_pair(v, 3, hold_e[v], hold_o[v]);
            }
#pragma unroll
            for (int v = 0; v < 7; ++v) split_pair(v, PSLOT >> 1, val[v], fsave[v]);
        }
    };
    // values: 0 a1 = 2 ebar n yhat, 1 ybar, 2 a2 = s ybar | 3 yhat, 4 yhat_osig, 5 u_k, 6 u_k_osig
    // The MFMAs of a finished octet (18, or 36 for BF16X3) are issued ONE (two) at a time at six points of each of the next
    // three steps: a v_mfma_f32_32x32x16_bf16 occupies the matrix pipe for 32 cycles, and an in-order wave that issues the
    // next one earlier simply waits (SQ_WAIT_INST_ANY was 118 cycles per step with three in a row).
    // The hooks are unconditional: before the first octet the fragments are zero, so its three leading steps apply nothing.
    bool pend = false;                // an octet has been recorded whose MFMAs are not all issued yet (decides the final flush)
#pragma unroll
    for (int v = 0; v < 7; ++v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { fH[v][r] = 0u; fL[v][r] = 0u; if constexpr (RANK1 == 2) fM[v][r] = 0u; }
    }
    constexpr int PER_PAIR = RANK1 == 2 ? 6 : 3, UNITS = 6 * PER_PAIR, PER_HOOK = RANK1 == 2 ? 2 : 1;
    typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
    // hook points of a step: A-points 0..5 (both modes) and B-points 0..5 (BF16X3 only): point q = 0..17 over the three steps
    // issues unit q (BF16X2) or unit 2 q at the A-point and unit 2 q + 1 at the B-point (BF16X3): never two MFMAs in a row
    auto mf_unit = [&](auto usel) {
        constexpr int U = decltype(usel)::value, PR = U / PER_PAIR, K = U % PER_PAIR;
        auto frag = [&](const unsigned (&f)[4]) { return __builtin_bit_cast(bf8, v4u{f[0], f[1], f[2], f[3]}); };
        auto piece = [&](auto which, int idx) -> const unsigned (&)[4] {        // 0 hi, 1 second 8 bits, 2 third 8 bits
            constexpr int W = decltype(which)::value;
            if constexpr (W == 0) return fH[idx];
            else if constexpr (W == 1) { if constexpr (RANK1 == 2) return fM[idx]; else return fL[idx]; }
            else return fL[idx];
        };
        // products of a pair, in the order they are issued: (hi,hi) (hi,2nd) (2nd,hi) | (hi,3rd) (3rd,hi) (2nd,2nd)
        constexpr int WA = K == 0 ? 0 : K == 1 ? 0 : K == 2 ? 1 : K == 3 ? 0 : K == 4 ? 2 : 1;
        constexpr int WB = K == 0 ? 0 : K == 1 ? 1 : K == 2 ? 0 : K == 3 ? 2 : K == 4 ? 0 : 1;
        auto mf = [&](v16f& acc, int ia, int ib) {
            if constexpr (RANK1 == 3)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(hf8, frag(piece(std::integral_constant<int, WA>{}, ia))),
                                                             __builtin_bit_cast(hf8, frag(piece(std::integral_constant<int, WB>{}, ib))), acc, 0, 0, 0);
            else
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag(piece(std::integral_constant<int, WA>{}, ia)),
                                                              frag(piece(std::integral_constant<int, WB>{}, ib)), acc, 0, 0, 0);
        };
        if constexpr (PR == 0) mf(Rre, 0, LEGACY ? 5 : 3);
        if constexpr (PR == 1) mf(Rim, 0, LEGACY ? 6 : 4);
        if constexpr (PR == 2) mf(Qre, 1, 5);
        if constexpr (PR == 3) mf(Qim, 1, 6);
        if constexpr (PR == 4) mf(Rre, 2, 5);
        if constexpr (PR == 5) mf(Rim, 2, 6);
    };
    auto mf_hook = [&](auto qsel, auto bsel) {    // hook q = 0 .. 17 (three steps x six points), A- or B-point
        constexpr int Q = decltype(qsel)::value;
        constexpr bool BP = decltype(bsel)::value;
        if constexpr (!BP || PER_HOOK == 2) {
            __builtin_amdgcn_sched_barrier(0);
            mf_unit(std::integral_constant<int, PER_HOOK * Q + (BP ? 1 : 0)>{});
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto flush_octet = [&]() {
#define FLUSH_Q(Q) mf_hook(std::integral_constant<int, (Q)>{}, std::false_type{}); mf_hook(std::integral_constant<int, (Q)>{}, std::true_type{});
        FLUSH_Q(0) FLUSH_Q(1) FLUSH_Q(2) FLUSH_Q(3) FLUSH_Q(4) FLUSH_Q(5) FLUSH_Q(6) FLUSH_Q(7) FLUSH_Q(8)
        FLUSH_Q(9) FLUSH_Q(10) FLUSH_Q(11) FLUSH_Q(12) FLUSH_Q(13) FLUSH_Q(14) FLUSH_Q(15) FLUSH_Q(16) FLUSH_Q(17)
#undef FLUSH_Q
        pend = false;
    };
    static_assert(UNITS == 18 * PER_HOOK, "every hook issues PER_HOOK MFMAs");
#define MF_HOOK_AB(H, BP)                                                                                      \
    if constexpr (decltype(slot)::value >= 5) {                                                                \
        mf_hook(std::integral_constant<int, (7 - (decltype(slot)::value >= 5 ? decltype(slot)::value : 7)) * 6 + (H)>{}, BP{}); \
    }
#define MF_HOOK(H) MF_HOOK_AB(H, std::false_type)
#define MF_HOOKB(H) MF_HOOK_AB(H, std::true_type)
    auto chain_step = [&](const Pre& S, float uk, float uko, auto have_pre, int jn, bool exact, auto slot) -> Pre {
        // ---- chain, scalar part ----
        MF_HOOK(0)
        facc += S.dtk * (go * S.un);
        const v2f yhbp = cmul2_conj_b(mk2(g, go), S.rho);              // conj(rho_k) g
        const float yhb = yhbp.x;
        float dot = rad_next;
        // Only the first step of a staged chunk projects explicitly, and inside an aligned octet that can only be slot 7: the other
        // slots carry no test at all, and the block stays a branch (sum64 is plain asm; the empty volatile asm keeps it from ever being
        // if-converted into five DPP adds per step).  Timing-neutral (A/B in one call, profiles/r5_c3_ab_projection_branch.log: hipcc
        // already branched around it) -- kept because it removes eight compare-and-branch pairs per octet.
        constexpr int SLOT = decltype(slot)::value;
        if constexpr (SLOT < 0 || SLOT == 7) {
            if (exact) {
                asm volatile("" ::: "memory");
                dot = LEGACY ? sum64(S.yhp * yhb) + rad_next : sum64(S.yhp * yhb);
            }
        }
        rad_next = S.rad;
        const float ybar = (yhb - dot * S.yhp) * S.inv + S.pre;
        bcast_issue(aBw, aBr, ybar, qc);                               // 9 ops
        MF_HOOKB(0)
        {   // M_k = Q + s_k R^dagger, in the shadow of the broadcast
            const v2f s2 = mk2(S.s, S.s);
#pragma unroll
            for (int m = 0; m < 8; ++m) MM[m] = __builtin_elementwise_fma(MRd[m], s2, MQ[m]);
            MF_HOOK(1)
#pragma unroll
            for (int m = 8; m < 16; ++m) MM[m] = __builtin_elementwise_fma(MRd[m], s2, MQ[m]);
        }
        MF_HOOKB(1)
        // ---- off-chain: pre of step k-1 (gives u_k) ----
        Pre Sn = S;
        if constexpr (decltype(have_pre)::value) {
            lds_wait_own<9>(yh_j, rho_j, c0_j, c1_j);
            Sn = make_pre(yh_j, rho_j, c0_j, c1_j);
            uk = Sn.un;
            uko = Sn.uno;
        }
        // ---- chain, mat-vec part: g_k = ybar + M_k ybar ----
        MF_HOOK(2)
        lds_wait_lo<4>(qc);
        v2f am;
        constexpr bool SPLIT_MV = RANK1 == 2 && decltype(slot)::value >= 5;   // BF16X3's B-points inside the mat-vec
        if constexpr (SPLIT_MV) {
            mv1_quarter<true>(MM[0], MM[1], MM[2], MM[3], qc[0], qc[1], am);
            MF_HOOKB(2)
            mv1_quarter<false>(MM[4], MM[5], MM[6], MM[7], qc[2], qc[3], am);
        } else {
            mv1_lo(MM, qc, am);
        }
        MF_HOOK(3)
        lds_wait_hi<0>(qc);
        if constexpr (SPLIT_MV) {
            mv1_quarter<false>(MM[8], MM[9], MM[10], MM[11], qc[4], qc[5], am);
            MF_HOOKB(3)
            mv1_quarter<false>(MM[12], MM[13], MM[14], MM[15], qc[6], qc[7], am);
        } else {
            mv1_hi(MM, qc, am);
        }
        MF_HOOK(4)
        const float md = swapadd_after_asm(am.x, am.y);
        accS += md * uk;
        g = ybar + md;
        MF_HOOKB(4)
        go = osig_of(g, hb);
        MF_HOOK(5)
        // ---- rank-1 gradient updates (A: rows i, B: columns j; K = {re, im}) ----
        //   Rbar += 2 ebar y y^dagger + s ybar u^dagger ;  Qbar += ybar u^dagger
        //   Re(a b^dagger): A = a (split), B = b (split);  Im(a b^dagger): A = a (split), B = -b_osig
        //   (the sign of the Im tiles is applied once at the end)
        const float a1 = LEGACY ? S.ten * uk : S.ten * S.yh;   // 2 ebar n yhat  (y y^dagger = n yhat yhat^dagger); legacy: te_k psi_k
        const float a2 = S.s * ybar;
        ybar_last = ybar;
        if constexpr (decltype(slot)::value < 0) {
            // (RANK1 == 3: the accumulators carry the scales sR 2^13 and sQ 2^13 -- exact powers of two on the operands of the exact MFMA)
            const float kA = RANK1 == 3 ? sR : 1.f, kQ = RANK1 == 3 ? sQ : 1.f, kB = RANK1 == 3 ? SB16 : 1.f;
            float o_a1 = a1 * kA, o_a2 = a2 * kA, o_yb = ybar * kQ, o_b1 = (LEGACY ? uk : S.yh) * kB, o_b1o = (LEGACY ? uko : S.yho) * kB;
            float o_uk = uk * kB, o_uko = uko * kB;
            // every operand is complete two wait states before the first MFMA reads it: hipcc places a v_pk_fma_f32 one instruction in
            // front of the v_mfma that reads its result (the matrix core then sees the previous step's u_k)
            asm volatile("s_nop 1" : "+v"(o_a1), "+v"(o_a2), "+v"(o_yb), "+v"(o_b1), "+v"(o_b1o), "+v"(o_uk), "+v"(o_uko));
            Rre = __builtin_amdgcn_mfma_f32_32x32x2f32(o_a1, o_b1, Rre, 0, 0, 0);
            Rim = __builtin_amdgcn_mfma_f32_32x32x2f32(o_a1, o_b1o, Rim, 0, 0, 0);
            Qre = __builtin_amdgcn_mfma_f32_32x32x2f32(o_yb, o_uk, Qre, 0, 0, 0);
            Qim = __builtin_amdgcn_mfma_f32_32x32x2f32(o_yb, o_uko, Qim, 0, 0, 0);
            Rre = __builtin_amdgcn_mfma_f32_32x32x2f32(o_a2, o_uk, Rre, 0, 0, 0);
            Rim = __builtin_amdgcn_mfma_f32_32x32x2f32(o_a2, o_uko, Rim, 0, 0, 0);
        } else {
            const float val[7] = {a1, ybar, a2, S.yh, S.yho, uk, uko};
            record(slot, val);
        }
        MF_HOOKB(5)
        if constexpr (decltype(slot)::value == 5) pend = false;
        return Sn;
    };

    // (a RhoCMPS column on its own has no radial identity -- Re(sum_a yhat_a^dagger g_a) = te e holds for the SUM over the clip's columns -- so
    // the explicit projection of a staged chunk's first step is a pure-state refinement: virtual clips use the analytic value throughout,
    // as the wide kernels do)
    const bool proj_ok = P.phi0 == nullptr;
    // pre index j runs N-2 .. 0, one staged chunk (32 steps) at a time; the chain handles step j+1 in the
    // same iteration; per-step scalars are re-derived whenever j enters a new 64-step chunk
    for (int hh = hl; hh >= 0; --hh) {
        const int jlo = hh * CHB;
        const int jhi = (N - 2) < (jlo + CHB - 1) ? (N - 2) : (jlo + CHB - 1);
        stage_load_all(hh > 0 ? hh - 1 : 0);          // prefetch into registers (clamped: harmless reload)
        const bool new_scal = (hh & 1) == 0 && hh > 0;
        if (new_scal) scal_load((hh >> 1) - 1);
        int j = jhi;
        // steps above the first aligned octet (top chunk only; EXACT_F32: every step): immediate fp32 updates
        for (; j >= jlo && (RANK1 == 0 || (j & 7) != 7); --j) {
            const int jr = j & (CHB - 1), jc = j & (CH - 1);
            own_issue(aYown + jr * 512, aRho + jr * 256, aScl + jc * 32, yh_j, rho_j, c0_j, c1_j);
            S = chain_step(S, 0.f, 0.f, std::true_type{}, j, proj_ok && j == jhi, std::integral_constant<int, -1>{});
        }
        // aligned octets: slot = j & 7, updates recorded and applied once per octet
#define BWD_STEP8(P)                                                                                          \
        {                                                                                                     \
            const int jq = j - (7 - (P));                                                                     \
            own_issue_off<(P) * 512, (P) * 256, (P) * 32>(aYo8, aRo8, aSo8, yh_j, rho_j, c0_j, c1_j);         \
            S = chain_step(S, 0.f, 0.f, std::true_type{}, jq, proj_ok && jq == jhi, std::integral_constant<int, (P)>{}); \
        }
        for (; j >= jlo; j -= 8) {
            if constexpr (RANK1 == 3) {
                // chain steps j + 1 .. j - 6 <-> positions (j & 63) - 7 .. (j & 63) of the chunk committed for loop indices j (scal_commit)
                const float Y = 1.001f * sqrtf(sum64(ybar_last * ybar_last));
                if (octet_bounds((j & (CH - 1)) & ~7, Y)) {
                    if (pend) flush_octet();                   // what is recorded goes in under the old scales ..
#pragma unroll
                    for (int v = 0; v < 7; ++v)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { fH[v][r] = 0u; fL[v][r] = 0u; }    // .. the unconditional hooks then add zeros
                    apply_scales();
                }
            }
            // rows of this octet in the staged chunk: j & 7 == 7 here, so row (j - 7 + P) = octet base + P
            const unsigned aYo8 = aYown + ((j - 7) & (CHB - 1)) * 512, aRo8 = aRho + ((j - 7) & (CHB - 1)) * 256;
            const unsigned aSo8 = aScl + ((j - 7) & (CH - 1)) * 32;
            BWD_STEP8(7) BWD_STEP8(6) BWD_STEP8(5) BWD_STEP8(4) BWD_STEP8(3) BWD_STEP8(2) BWD_STEP8(1) BWD_STEP8(0)
            pend = true;                               // applied during the next step (or by the final flush)
        }
#undef BWD_STEP8
        if (hh > 0) stage_commit_all();
        if (new_scal) {
            scal_commit((hh >> 1) - 1);
        }
    }
    if (pend) flush_octet();
    S = chain_step(S, u0, u0o, std::false_type{}, 0, proj_ok, std::integral_constant<int, -1>{});   // step 0: u_0 = psi_0
#undef MF_HOOK
#undef MF_HOOKB
#undef MF_HOOK_AB

    // ---------------- per-clip slab ----------------
    float* slab = P.slabs + (size_t)b * P.slab_floats;
    constexpr int DD = DPW * DPW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;   // C/D layout of the 32x32 MFMA: column = lane & 31
        const int o = row * DPW + i;
        const float uR = RANK1 == 3 ? pow2_inv(sR) * (1.0f / SB16) : 1.f, uQ = RANK1 == 3 ? pow2_inv(sQ) * (1.0f / SB16) : 1.f;
        slab[o] = Rre[r] * uR;
        slab[DD + o] = -Rim[r] * uR;
        slab[2 * DD + o] = Qre[r] * uQ;
        slab[3 * DD + o] = -Qim[r] * uQ;
    }
    const float sumS = sum64(accS);
    const float sumA = av == 0 ? sum64(accA) : 0.f;        // (z = e x / A belongs to the clip: counted with its first column)
    // fbar_i = sum_k dtk Im(g conj(u)) = sum over both halves of g_osig * u_own (the sign lives in osig)
    const float ftot = swapadd(facc, facc);               // half 0: f(h=0) + f(h=1)
    slab[4 * DD + (hb ? 2 * DPW : DPW) + i] = g;          // cotangent of psi_0: re in [DPW, 2DPW), im in [2DPW, 3DPW)
    if (!hb) slab[4 * DD + i] = ftot;
    if (lane == 0) {
        // Abar = sum_k zbar_k (-(e x)_k / A^2) + sum_k sbar_k (-x_k / A^2),  s_k = x_k / A;  sumS still contains
        // sum_k Re(u_k^dagger Q ybar_k), which k_finalize removes (Dev::abar_fix)
        slab[4 * DD + 3 * DPW] = -(sumA / (A * A)) - sumS / A;
        slab[4 * DD + 3 * DPW + 1] = 0.f;
    }
}


// ------------------------------------------------------------------------------------------------
// sampling: PsiCMPS.sample = tf.scan of _psi_and_sample_update (model.py:242-251, 284-291)
//   increment = 2 Re(u^dagger R u) * delta_t + noise_k;  sample += increment;  psi <- update(psi, increment);  normalise
// One wavefront per sample path, same layout and mat-vec as the forward scan; here both wave reductions sit on
// the serial chain (the increment feeds the update), so this kernel is latency-bound by construction.
// ------------------------------------------------------------------------------------------------
namespace {

// Two mat-vec chains (this half's 16 columns of MA u and MB u) with the wave reduction of `x` threaded through them (one DPP
// step every four packed FMAs, as mv1r_lo / mv1r_hi of cmps_wave2.hip); after the second block `tot` (SGPR) = sum of x over
// the 64 lanes.
#define SDPP(ctrl) "v_add_f32_dpp %[x], %[x], %[x] " ctrl " row_mask:0xf bank_mask:0xf\n\t"
__device__ __forceinline__ void mv2r_lo(const v2f (&MA)[16], const v2f (&MB)[16], const v4f (&q)[8], v2f& accA, v2f& accB, float& x) {
    asm(CM_FIRST([a], [a0], [q0]) CM_FIRST([b], [b0], [q0]) SDPP("quad_perm:[1,0,3,2]")
        CM([a], [a1], [q1]) CM([b], [b1], [q1]) CM([a], [a2], [q2]) CM([b], [b2], [q2]) SDPP("quad_perm:[2,3,0,1]")
        CM([a], [a3], [q3]) CM([b], [b3], [q3]) CM([a], [a4], [q4]) CM([b], [b4], [q4]) SDPP("row_half_mirror")
        CM([a], [a5], [q5]) CM([b], [b5], [q5]) CM([a], [a6], [q6]) CM([b], [b6], [q6]) SDPP("row_mirror")
        CM([a], [a7], [q7]) CM([b], [b7], [q7])
        : [a] "=&v"(accA), [b] "=&v"(accB), [x] "+v"(x)
        : [a0] "v"(MA[0]), [a1] "v"(MA[1]), [a2] "v"(MA[2]), [a3] "v"(MA[3]), [a4] "v"(MA[4]), [a5] "v"(MA[5]), [a6] "v"(MA[6]), [a7] "v"(MA[7]),
          [b0] "v"(MB[0]), [b1] "v"(MB[1]), [b2] "v"(MB[2]), [b3] "v"(MB[3]), [b4] "v"(MB[4]), [b5] "v"(MB[5]), [b6] "v"(MB[6]), [b7] "v"(MB[7]),
          [q0] "v"(lo2(q[0])), [q1] "v"(hi2(q[0])), [q2] "v"(lo2(q[1])), [q3] "v"(hi2(q[1])), [q4] "v"(lo2(q[2])), [q5] "v"(hi2(q[2])),
          [q6] "v"(lo2(q[3])), [q7] "v"(hi2(q[3])));
}
__device__ __forceinline__ void mv2r_hi(const v2f (&MA)[16], const v2f (&MB)[16], const v4f (&q)[8], v2f& accA, v2f& accB, float& x,
                                        float& tot) {
    asm(CM([a], [a0], [q0]) CM([b], [b0], [q0]) CM([a], [a1], [q1]) CM([b], [b1], [q1])
        "v_add_f32_dpp %[x], %[x], %[x] row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        CM([a], [a2], [q2]) CM([b], [b2], [q2]) CM([a], [a3], [q3]) CM([b], [b3], [q3])
        "v_add_f32_dpp %[x], %[x], %[x] row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        CM([a], [a4], [q4]) CM([b], [b4], [q4]) CM([a], [a5], [q5]) CM([b], [b5], [q5])
        "v_readlane_b32 %[t], %[x], 63\n\t"
        CM([a], [a6], [q6]) CM([b], [b6], [q6]) CM([a], [a7], [q7]) CM([b], [b7], [q7])
        : [a] "+v"(accA), [b] "+v"(accB), [x] "+v"(x), [t] "=s"(tot)
        : [a0] "v"(MA[8]), [a1] "v"(MA[9]), [a2] "v"(MA[10]), [a3] "v"(MA[11]), [a4] "v"(MA[12]), [a5] "v"(MA[13]), [a6] "v"(MA[14]),
          [a7] "v"(MA[15]),
          [b0] "v"(MB[8]), [b1] "v"(MB[9]), [b2] "v"(MB[10]), [b3] "v"(MB[11]), [b4] "v"(MB[12]), [b5] "v"(MB[13]), [b6] "v"(MB[14]),
          [b7] "v"(MB[15]),
          [q0] "v"(lo2(q[4])), [q1] "v"(hi2(q[4])), [q2] "v"(lo2(q[5])), [q3] "v"(hi2(q[5])), [q4] "v"(lo2(q[6])), [q5] "v"(hi2(q[6])),
          [q6] "v"(lo2(q[7])), [q7] "v"(hi2(q[7])));
}
#undef SDPP

}  // namespace

// Round 3: the normalisation is linear, so the wave carries ut = rho_{k-1} y_{k-1} UN-normalised and the reduction of |y_{k-1}|^2 rides
// inside the FMA blocks of step k (as in the forward scan, cmps_wave2.hip): with inv = rsqrt(max(|y_{k-1}|^2, 1e-12)),
//   e = 2 inv^2 Re(ut^dagger R ut),   y_k = inv (ut + Q ut + s R ut),
// and only the expectation's reduction is left on the serial chain.
__global__ __launch_bounds__(64 * WAVES, 1) void k_sample_wave(Dev P, const float* __restrict__ noise, int n_paths,
                                                               int length, float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float4 stR[WAVES][CH * 16];
    __shared__ __attribute__((aligned(16))) float2 bcU[WAVES][DPW];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    const int b = blockIdx.x * WAVES + w;
    if (b >= n_paths) return;
    const int N = length, NC = (N + CH - 1) / CH;
    v2f MR[16], MQ[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        MR[m] = ld2(&P.R[i * DPW + 16 * h + m]);
        MQ[m] = ld2(&P.Q[i * DPW + 16 * h + m]);
    }
    const unsigned aUw = lds_addr(&bcU[w][0]) + i * 8 + h * 4, aUr = lds_addr(&bcU[w][0]) + h * 128;
    const unsigned aRho = lds_addr(&stR[w][0]) + i * 8;
    const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
    const float* nrow = noise + (size_t)b * length;
    float* orow = out + (size_t)b * length;
    const float A = dev_A(P), dt = P.dt;
    const float2 p0 = P.psi0[i];
    float u = hb ? p0.y : p0.x;
    float xsq = lane == 0 ? 1.f : 0.f;                   // "|y_{-1}|^2" = 1: psi_0 arrives normalised
    float samp = 0.f;                                    // model.py:244 batch_zeros
    v4f sr[16], qu[8];
    v2f rho;
    for (int c = 0; c < NC; ++c) {
        const int kbeg = c * CH;
        const int cnt = (N - kbeg) < CH ? (N - kbeg) : CH;
        stage_load<16>(rho4, kbeg, P.N, lane, sr);
        const float nz = kbeg + lane < N ? nrow[kbeg + lane] : 0.f;
        stage_commit<16>(stR[w], lane, sr);
        float svec = 0.f;
        for (int kk = 0; kk < cnt; ++kk) {
            bcast_issue_tab(aUw, aUr, u, aRho + kk * 256, qu, rho);
            lds_wait_lo<5>(qu);
            v2f av, aq;
            float nprev;
            mv2r_lo(MR, MQ, qu, av, aq, xsq);
            lds_wait_hi_t<0>(qu, rho);
            mv2r_hi(MR, MQ, qu, av, aq, xsq, nprev);                 // nprev = |y_{k-1}|^2
            const float vs = swapadd(av.x, av.y), qs = swapadd(aq.x, aq.y);
            const float inv = rsq_nr(fmaxf(nprev, 1e-12f));          // :289 of the step before
            const float e = 2.0f * (sum64(u * vs) * inv) * inv;      // _expectation on the normalised state (model.py:319-325)
            const float inc = e * dt + rdlane(nz, kk);               // model.py:286
            samp += inc;                                             // :287
            svec = (lane == kk) ? samp : svec;
            const float s = inc / A;                                 // :288 -> :303
            const float y = inv * (u + (qs + s * vs));
            const float yo = osig_of(y, hb);
            const v2f un = cmul2(mk2(y, yo), rho);                   // rho_k y_k, normalised in the next step
            u = un.x;
            xsq = y * y;
        }
        if (lane < cnt) orow[kbeg + lane] = A * svec;                // model.py:251
    }
}

hipError_t launch_sample_wave(const Dev& P, const float* noise, int n, int length, float* out, hipStream_t s) {
    const unsigned nb = (unsigned)((n + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(k_sample_wave, dim3(nb), dim3(64 * WAVES), 0, s, P, noise, n, length, out);
    return hipGetLastError();
}

hipError_t launch_bwd_legacy_wave(const Dev& P, const float* audio, int rank1_mode, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (rank1_mode == 0)
        hipLaunchKernelGGL((k_bwd_wave<0, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    else if (rank1_mode == 1)
        hipLaunchKernelGGL((k_bwd_wave<1, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    else
        hipLaunchKernelGGL((k_bwd_wave<2, true>), dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    return hipGetLastError();
}

hipError_t launch_bwd_wave(const Dev& P, const float* audio, int rank1_mode, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    if (rank1_mode == 0)
        hipLaunchKernelGGL(k_bwd_wave<0>, dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    else if (rank1_mode == 1)
        hipLaunchKernelGGL(k_bwd_wave<1>, dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    else if (rank1_mode == 3)
        hipLaunchKernelGGL(k_bwd_wave<3>, dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    else
        hipLaunchKernelGGL(k_bwd_wave<2>, dim3(nb), dim3(64 * WAVES), 0, s, P, audio);
    return hipGetLastError();
}

}  // namespace cmps
