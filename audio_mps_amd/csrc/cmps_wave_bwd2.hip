// Reverse scan of the pure-state model for 16 < D <= 32 with TWO wavefronts per clip on one SIMD (round 5) -- the forward's recipe
// (cmps_wave2.hip) applied to the reverse sweep.
//
// Why: a lone wave issues one instruction per ~5.4-5.8 cycles, two waves on one SIMD one per ~4.3 combined, and at B = 1024 clips
// there is exactly one clip per SIMD.  k_bwd_wave<F16X2> (cmps_wave.hip) issues 135 instructions per step from ONE wave
// (profiles/r5_c3_isa_budget.log); 19 of them split the seven rank-1 operand values into fp16 pieces, 2.25 are the MFMAs, ~8 move
// values around for them, ~4 maintain the guaranteed bounds the scales follow.  Nothing on the serial chain waits for any of that:
//
//   chain wave  (waves 0-3 of the workgroup, raised priority): the recurrence g -> conj(rho) g -> ybar -> broadcast -> M_k ybar -> g
//     exactly as in k_bwd_wave, the off-chain "pre" stage one step ahead, the staging of the stash / rho / scalar rows, the f / psi_0 /
//     A parts of the slab.  Per step it leaves THREE floats per lane in an LDS ring: ybar_k, yhat_k, u_k (split layout).
//   gradient wave  (waves 4-7): an octet (eight steps) at a time it reads the ring (the partner component of a value is just another LDS
//     address: no lane exchange), forms a1 = ten_k yhat_k and a2 = s_k ybar_k with per-step scalars it derives itself from the audio and
//     the forward's scalar rows, picks the fp16 scales of the octet from the MEASURED maxima of the values it is about to split (it has
//     them all in hand, so no bound has to be guaranteed in advance: the round-4 suffix scans of affine maps are gone), splits, and
//     issues the 18 v_mfma_f32_32x32x16_f16 of the octet; at the end it writes the R / Q sections of the slab.
//
// Synchronisation as in the forward: two LDS counters per clip (octets produced / consumed), no barrier; the ring holds two octets.
// Every wait loop is bounded (a wave that never sees its partner's counter gives up and finishes: wrong numbers, never a hang).
// The first octet is padded in FRONT with zeros when (N - 1) & 7 steps sit above the first aligned octet, step 0 gets an octet of its own
// (seven zero slots behind it): zero operands add nothing.
// Arithmetic: the chain is k_bwd_wave's instruction for instruction; the sums are the F16X2 form (three products per pair, fp32
// accumulate), with scales that differ from k_bwd_wave<3>'s by powers of two only where both are in range -- same accuracy class
// (tests/test_gpu_parity.py::test_two_wave_reverse_scan_*).  PsiCMPS arithmetic only (no legacy mode); RhoCMPS virtual clips are supported.
#include "cmps_wave_util.h"

namespace cmps {

namespace {

struct Pre2 {
    float yh, yho, yhp, un, pre;
    v2f rho;
    v2f sd;                                    // (s_k, dt_k): kept as a pair so that v_pk_fma can read s from its low half
    float inv, ten, rad;
};

constexpr int RING_SLOT = 256;                 // bytes: one value of one step, 64 lanes
constexpr int RING_VAL = 8 * RING_SLOT;        // one value, eight steps
constexpr int RING_HALF = 3 * RING_VAL;        // (ybar, yhat, u) of one octet
constexpr int SPIN_MAX = 1 << 22;              // bound of every wait loop
constexpr int MT = 4;                          // entries of M_{k-1} formed in the tail of step k (chain_step: inside its two lane exchanges)

__device__ __forceinline__ int flag_load2(unsigned addr) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void flag_store2(unsigned addr_l, int v) {       // addr_l: the flag in lane 0, a sink word in the other lanes
    asm volatile("ds_write_b32 %0, %1" : : "v"(addr_l), "v"(v) : "memory");
}
// v_permlane32_swap_b32 a, b (a's upper half <-> b's lower half) needs two wait states behind the VALU writes of a and b; hipcc fills
// them with s_nop 1 although independent work is at hand.  Here they hold two entries of the next step's matrix M = Q + s R^dagger.
__device__ __forceinline__ void swap_fill(float& a, float& b, v2f& m0, v2f r0, v2f q0, v2f& m1, v2f r1, v2f q1, v2f s2) {
    asm("v_pk_fma_f32 %2, %4, %8, %5 op_sel_hi:[1,0,1]\n\t"
        "v_pk_fma_f32 %3, %6, %8, %7 op_sel_hi:[1,0,1]\n\t"
        "v_permlane32_swap_b32 %0, %1"
        : "+v"(a), "+v"(b), "=&v"(m0), "=&v"(m1) : "v"(r0), "v"(q0), "v"(r1), "v"(q1), "v"(s2));
}
template <int OFF>
__device__ __forceinline__ void ring_write(unsigned addr, float v) {
    asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ float wave_max(float x) {       // max over the 64 lanes (x >= 0), uniform; DPP as sum64 (cmps_wave_util.h)
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf" : "+v"(x));
    return rdlane(x, 63);
}

}  // namespace

__global__ __launch_bounds__(128 * WAVES, 1) void k_bwd_wave2w(Dev P, const float* __restrict__ audio) {
    __shared__ __attribute__((aligned(16))) float4 stY[WAVES][CHB * 32];   // stashed (y, H y) rows of the staged chunk
    __shared__ __attribute__((aligned(16))) float4 stR[WAVES][CHB * 16];   // rho rows
    __shared__ __attribute__((aligned(16))) float4 scl[WAVES][CH * 2];     // per-step scalars, one 32-B row per step
    __shared__ __attribute__((aligned(16))) float2 bcB[WAVES][DPW];
    __shared__ __attribute__((aligned(16))) float ring[WAVES][2 * 3 * 8 * 64];   // [half][ybar | yhat | u][slot][lane]
    __shared__ __attribute__((aligned(8))) float2 sct[WAVES][2][CH];             // (s_k, ten_k) of a 64-step chunk, by chunk parity: for the gradient wave
    __shared__ int flags[WAVES][2];                                         // [clip][0: octets produced, 1: octets consumed]
    __shared__ int flag_sink[WAVES][2][64];                                 // where lanes 1 .. 63 of a flag store write (no exec masking in the loops)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int w = wv & (WAVES - 1), role = wv / WAVES;
    const int i = lane & 31, h = lane >> 5;
    const bool hb = h != 0;
    if (threadIdx.x < 2 * WAVES) (&flags[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const int b = blockIdx.x * WAVES + w;      // wave-uniform
    if (b >= P.B) return;                      // both waves of the clip leave together
    const int N = P.N, T = P.T, NC = (N + CH - 1) / CH;
    // hand-over order: the (N - 1) & 7 steps above the first aligned octet, the aligned octets, step 0
    const int u_top = (N - 1) & 7, a_oct = (N - 1 - u_top) >> 3, n_oct = (u_top ? 1 : 0) + a_oct + 1;
    const unsigned aProd = lds_addr(&flags[w][0]), aCons = lds_addr(&flags[w][1]);
    const unsigned aProdL = lane == 0 ? aProd : lds_addr(&flag_sink[w][0][lane]), aConsL = lane == 0 ? aCons : lds_addr(&flag_sink[w][1][lane]);
    const unsigned aRing0 = lds_addr(&ring[w][0]) + lane * 4;
    // RhoCMPS on virtual clips (Dev::phi0, cmps_rho_wave.hip): wave pair b is column av of clip bc
    const int vr = P.phi0 ? P.phi_rank : 1;
    const int bc = b / vr, av = b - bc * vr;
    const float* xrow = audio + (size_t)bc * T;
    const float* sc = P.scal + scal_off(bc, NC, 0);
    const float A = dev_A(P);
    float* slab = P.slabs + (size_t)b * P.slab_floats;
    constexpr int DD = DPW * DPW;

    if (role == 0) {
        // ================================================================== chain wave
        __builtin_amdgcn_s_setprio(3);
        stagger(w);
        v2f MRd[16], MQ[16], MM[16];
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const v2f rt = ld2(&P.RT[i * DPW + 16 * h + m]);   // R[16h+m][i]
            MRd[m] = mk2(rt.x, -rt.y);                          // R^dagger[i][16h+m]
            MQ[m] = ld2(&P.Q[i * DPW + 16 * h + m]);
        }
        const unsigned aBw = lds_addr(&bcB[w][0]) + i * 8 + h * 4, aBr = lds_addr(&bcB[w][0]) + h * 128;
        const unsigned aYown = lds_addr(&stY[w][0]) + (2 * i + h) * 8;   // stash rows: 64 (y[n], (H y)[n]) pairs, n = 2 i + {re, im}
        const unsigned aRho = lds_addr(&stR[w][0]) + i * 8;
        const unsigned aScl = lds_addr(&scl[w][0]);
        const float4* rho4 = reinterpret_cast<const float4*>(P.rho);
        const float4* sty4 = reinterpret_cast<const float4*>(P.hst + ((size_t)bc * N * vr + av) * 128);
        float facc = 0.f, accS = 0.f, accA = 0.f;
        float ra0 = 0.f, ra1 = 0.f, rdt = 0.f, rnv = 1.f, rev = 0.f;   // raw prefetched values of the next chunk
        // staging by OCTETS (two waves share the SIMD's registers: 256 each): while the chain works through rows 8 q .. 8 q + 7 of the staged
        // chunk, the same eight rows of the chunk below wait in 24 registers and take over the slots when the octet is done
        v4f so4[4], so2[2];
        auto octet_load = [&](int hh, int q) {
            stage_load512<4>(sty4, hh * CHB + 8 * q, N - 1, lane, so4, vr);
            stage_load<2>(rho4, hh * CHB + 8 * q, N, lane, so2);
        };
        // A chunk below the top one (every row exists).  PsiCMPS stash (stride 1): the six loads as asm with a scalar base and constant
        // offsets -- written in C++ their address registers share the destination registers of the previous octet's loads and hipcc puts
        // an s_waitcnt vmcnt(0) BETWEEN the loads (a full memory latency per octet: 5.80 -> 6.24 ms, profiles/r5_c3_ab_two_wave.log).  The
        // compiler does not count asm loads: octet_commit_below waits for them itself.
        const unsigned so_lane16 = (unsigned)lane * 16u;
        auto octet_load_below = [&](int hh, int q) {
            if (vr == 1) {
                const int row0 = __builtin_amdgcn_readfirstlane(hh * CHB + 8 * q);
                const float4* py = sty4 + (size_t)row0 * 32;
                const float4* pr = rho4 + (size_t)row0 * 16;
                asm volatile("global_load_dwordx4 %0, %6, %7\n\tglobal_load_dwordx4 %1, %6, %7 offset:1024\n\t"
                             "global_load_dwordx4 %2, %6, %7 offset:2048\n\tglobal_load_dwordx4 %3, %6, %7 offset:3072\n\t"
                             "global_load_dwordx4 %4, %6, %8\n\tglobal_load_dwordx4 %5, %6, %8 offset:1024"
                             : "=&v"(so4[0]), "=&v"(so4[1]), "=&v"(so4[2]), "=&v"(so4[3]), "=&v"(so2[0]), "=&v"(so2[1])
                             : "v"(so_lane16), "s"(py), "s"(pr) : "memory");
            } else {
                octet_load(hh, q);
            }
        };
        auto octet_commit = [&](int q) {
            stage_commit<4>(&stY[w][q * 256], lane, so4);
            stage_commit<2>(&stR[w][q * 128], lane, so2);
        };
        auto octet_commit_below = [&](int q) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(so4[0]), "+v"(so4[1]), "+v"(so4[2]), "+v"(so4[3]), "+v"(so2[0]), "+v"(so2[1]) : : "memory");
            octet_commit(q);
        };
        auto scal_load = [&](int c) {
            const int idx = c * CH + lane;
            ra0 = idx < T ? xrow[idx] : 0.f;
            ra1 = idx + 1 < T ? xrow[idx + 1] : 0.f;
            rdt = P.dtk[idx];                 // padded to N + 64 entries
            rnv = sc[(size_t)c * 128 + lane];
            rev = sc[(size_t)c * 128 + 64 + lane];
        };
        auto scal_commit = [&](int c) {
            const int idx = c * CH + lane;
            const float inc = ra1 - ra0;
            const float nv = rnv, ev = rev;
            const float invv = rsq_nr(fmaxf(nv, 1e-12f));
            const float invokv = nv > 1e-12f ? invv : 0.f;
            const float sv = inc / A;
            const float ex = ev * inc;                      // model.py:294 operation order
            const float z = ex / A;
            const float zbar = -1.0f / (1.0f + z);
            const float ebar = zbar * inc / A;
            const float tev = 2.0f * ebar;
            scl[w][2 * lane] = make_float4(sv, rdt, invv, tev * nv);
            scl[w][2 * lane + 1] = make_float4(tev, invokv, tev * ev, 0.f);
            sct[w][c & 1][lane] = make_float2(sv, tev * nv);      // what the gradient wave needs of the row (it may lag two octets behind)
            if (idx < N) accA += zbar * ex;
        };
        // (sfill: the pair whose low half is s of the step whose matrix entries 10, 11 ride in the lane exchange: the CURRENT step's)
        auto make_pre = [&](v2f yh2, v2f rho, v4f c0, v4f c1, v2f sfill) -> Pre2 {
            const float yown = yh2.x, hown = yh2.y;
            Pre2 S;
            S.rho = rho;
            S.sd = __builtin_shufflevector(c0, c0, 0, 1);
            S.inv = c0.z;
            S.ten = c0.w;
            const float te = c1.x;
            const float invok = c1.y;
            S.rad = c1.z;
            S.pre = te * hown;
            S.yh = S.inv * yown;
            S.yhp = invok * yown;
            {
                float ya = S.yh, yb = S.yh;                                   // osig_of
                swap_fill(ya, yb, MM[10], MRd[10], MQ[10], MM[11], MRd[11], MQ[11], sfill);
                S.yho = hb ? -ya : yb;
            }
            const v2f un = cmul2(mk2(S.yh, S.yho), rho);
            S.un = un.x;
            return S;
        };

        const int hl = (N - 1) / CHB;
#pragma unroll 1
        for (int q = 0; q < 4; ++q) { octet_load(hl, q); octet_commit(q); }      // the top chunk (row (N - 1) & 31 feeds the first pre stage)
        scal_load(hl >> 1);
        scal_commit(hl >> 1);
        v4f qc[8];
        v2f yh_j, rho_j;
        v4f c0_j, c1_j;
        Pre2 S;
        {
            const int jr = (N - 1) & (CHB - 1), jc = (N - 1) & (CH - 1);
            own_issue(aYown + jr * 512, aRho + jr * 256, aScl + jc * 32, yh_j, rho_j, c0_j, c1_j);
            lds_wait_own<0>(yh_j, rho_j, c0_j, c1_j);
            S = make_pre(yh_j, rho_j, c0_j, c1_j, __builtin_shufflevector(c0_j, c0_j, 0, 1));    // (first step: its own row)
        }
        {   // the first step's tail-formed matrix entries (chain_step)
            const v2f s2 = mk2(S.sd.x, S.sd.x);
#pragma unroll
            for (int m = 16 - MT; m < 16; ++m) MM[m] = __builtin_elementwise_fma(MRd[m], s2, MQ[m]);
        }
        float g = 0.f, go = 0.f;                      // cotangent of u_{k+1}: split value and its osig
        const float2 p0 = P.phi0 ? P.phi0[av * DPW + i] : P.psi0[i];
        const float u0 = hb ? p0.y : p0.x;
        float rad_next = 0.f;                         // no step N: g_N = 0

        // one step of the serial chain (k_bwd_wave's, cmps_wave.hip: the derivation is there); the step's three operand vectors go to
        // ring slot `aslot` (an LDS address with the value / slot offsets folded in by the caller)
        auto chain_step = [&](const Pre2& S, float uk, auto have_pre, bool exact, unsigned aslot, auto slot_off) -> Pre2 {
            constexpr int SO = decltype(slot_off)::value;        // constant part of the slot address (the aligned octets: the whole slot)
            facc += S.sd.y * (go * S.un);
            const v2f yhbp = cmul2_conj_b(mk2(g, go), S.rho);              // conj(rho_k) g
            const float yhb = yhbp.x;
            float dot = rad_next;
            if (exact) {                                                   // (a real branch: see k_bwd_wave)
                asm volatile("" ::: "memory");
                dot = sum64(S.yhp * yhb);
            }
            rad_next = S.rad;
            const float ybar = (yhb - dot * S.yhp) * S.inv + S.pre;
            bcast_issue(aBw, aBr, ybar, qc);                               // 9 ops
            {   // M_k = Q + s_k R^dagger, in the shadow of the broadcast (entries 10, 11: inside this step's pre stage, make_pre; the last MT: in
                // the tail of the step before)
                const v2f s2 = mk2(S.sd.x, S.sd.x);
#pragma unroll
                for (int m = 0; m < 16 - MT - 2; ++m) MM[m] = __builtin_elementwise_fma(MRd[m], s2, MQ[m]);
            }
            Pre2 Sn = S;
            if constexpr (!decltype(have_pre)::value) {                     // no pre stage to carry entries 10, 11 (step 0)
                const v2f s2 = mk2(S.sd.x, S.sd.x);
                MM[10] = __builtin_elementwise_fma(MRd[10], s2, MQ[10]);
                MM[11] = __builtin_elementwise_fma(MRd[11], s2, MQ[11]);
            }
            if constexpr (decltype(have_pre)::value) {
                lds_wait_own<9>(yh_j, rho_j, c0_j, c1_j);
                Sn = make_pre(yh_j, rho_j, c0_j, c1_j, S.sd);
                uk = Sn.un;
            }
            // the gradient wave's operands of this step (behind the broadcast: the counted waits below only get stricter)
            ring_write<SO>(aslot, ybar);
            ring_write<SO + RING_VAL>(aslot, S.yh);
            ring_write<SO + 2 * RING_VAL>(aslot, uk);
            lds_wait_lo<4 + 3>(qc);
            v2f am;
            mv1_lo(MM, qc, am);
            lds_wait_hi<3>(qc);
            mv1_hi(MM, qc, am);
            // the tail is one dependency chain with two lane exchanges in it (VALU write -> permlane read: wait states the compiler fills
            // with s_nop): MT entries of the NEXT step's matrix go there (the mat-vec above has consumed this step's)
            const v2f sn2 = __builtin_shufflevector(c0_j, c0_j, 0, 1);     // (s, dtk) of the next step's row: only the low half is read
            float sx = am.x, sy = am.y;                                    // swapadd (cmps_wave_util.h)
            swap_fill(sx, sy, MM[12], MRd[12], MQ[12], MM[13], MRd[13], MQ[13], sn2);
            const float md = sx + sy;
            accS += md * uk;
            g = ybar + md;
            float ga = g, gb = g;                                          // osig_of
            swap_fill(ga, gb, MM[14], MRd[14], MQ[14], MM[15], MRd[15], MQ[15], sn2);
            go = hb ? -ga : gb;
            return Sn;
        };
        // hand an octet over: (wait until the gradient wave has left this half: octets o - 2 and below consumed) ... written ... publish
        int oct = 0;
        auto half_base = [&](int o) { return aRing0 + (unsigned)(o & 1) * RING_HALF; };
        int cons_seen = 0;                             // the consumer count as read in the middle of the previous octet (never ahead of the truth)
        auto peek_cons = [&]() {                       // issued in front of a step's own reads: the step's counted waits cover it
            asm volatile("ds_read_b32 %0, %1" : "=v"(cons_seen) : "v"(aCons) : "memory");
        };
        auto wait_free = [&](int o) {
            // (the register is valid only behind the counted waits of the steps since peek_cons: a volatile asm keeps its place among them,
            // the readfirstlane builtin could be scheduled right behind the read)
            int seen;
            asm volatile("s_nop 0\n\tv_readfirstlane_b32 %0, %1" : "=s"(seen) : "v"(cons_seen));
            if (o >= 2 && seen < o - 1) {
                int spin = 0;
                while (flag_load2(aCons) < o - 1 && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(1);
            }
        };
        auto publish = [&]() {
            ++oct;
            flag_store2(aProdL, oct);
        };

        const bool proj_ok = P.phi0 == nullptr;
        // octets of the top chunk that hold no loop index (rows above N - 2) never come up for a refill: the chunk below moves in now
        if (hl >= 1) {
            const int jtop = N - 2, q_hi = jtop >= hl * CHB ? (jtop & (CHB - 1)) >> 3 : -1;
#pragma unroll 1
            for (int q = q_hi + 1; q < 4; ++q) { octet_load(hl - 1, q); octet_commit(q); }
        }
        if (u_top) {                                   // zero slots in front of the first, partial octet
            const unsigned ab = half_base(0);
            for (int t = 0; t < 8 - u_top; ++t) {
                asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:%2\n\tds_write_b32 %0, %1 offset:%3"
                             : : "v"(ab + t * RING_SLOT), "v"(0.f), "n"(RING_VAL), "n"(2 * RING_VAL) : "memory");
            }
        }
        for (int hh = hl; hh >= 0; --hh) {
            const int jlo = hh * CHB;
            const int jhi = (N - 2) < (jlo + CHB - 1) ? (N - 2) : (jlo + CHB - 1);
            const bool new_scal = (hh & 1) == 0 && hh > 0;
            if (new_scal) scal_load((hh >> 1) - 1);
            int j = jhi;
            // steps above the first aligned octet (top chunk only): slots 8 - u_top .. 7 of octet 0; they are one whole octet of rows
            if (j >= jlo && (j & 7) != 7) {
                const unsigned ab = half_base(0);
                const int q = (j & (CHB - 1)) >> 3;
                if (hh > 0) octet_load(hh - 1, q);
                int t = 8 - u_top;
                for (; j >= jlo && (j & 7) != 7; --j, ++t) {
                    const int jr = j & (CHB - 1), jc = j & (CH - 1);
                    own_issue(aYown + jr * 512, aRho + jr * 256, aScl + jc * 32, yh_j, rho_j, c0_j, c1_j);
                    S = chain_step(S, 0.f, std::true_type{}, proj_ok && j == jhi, ab + t * RING_SLOT, std::integral_constant<int, 0>{});
                }
                if (hh > 0) octet_commit(q);
                publish();
            }
            // aligned octets: loop index j - 7 + P  <->  ring slot 7 - P
#define BWD2_STEP8(PQ)                                                                                         \
            {                                                                                                 \
                if ((PQ) == 3) peek_cons();                                                                                \
                own_issue_off<(PQ) * 512, (PQ) * 256, (PQ) * 32>(aYo8, aRo8, aSo8, yh_j, rho_j, c0_j, c1_j);  \
                S = chain_step(S, 0.f, std::true_type{}, (PQ) == 7 && proj_ok && j == jhi, ab8, std::integral_constant<int, (7 - (PQ)) * RING_SLOT>{}); \
            }
            for (; j >= jlo; j -= 8) {
                const int q = (j & (CHB - 1)) >> 3;
                if (hh > 0) octet_load_below(hh - 1, q);  // the chunk below, these eight rows: requested now, committed behind the octet
                wait_free(oct);
                const unsigned ab8 = half_base(oct);
                const unsigned aYo8 = aYown + ((j - 7) & (CHB - 1)) * 512, aRo8 = aRho + ((j - 7) & (CHB - 1)) * 256;
                const unsigned aSo8 = aScl + ((j - 7) & (CH - 1)) * 32;
                BWD2_STEP8(7) BWD2_STEP8(6) BWD2_STEP8(5) BWD2_STEP8(4) BWD2_STEP8(3) BWD2_STEP8(2) BWD2_STEP8(1) BWD2_STEP8(0)
                if (hh > 0) octet_commit_below(q);
                publish();
            }
#undef BWD2_STEP8
            if (new_scal) scal_commit((hh >> 1) - 1);
        }
        {   // step 0 (u_0 = psi_0): slot 0 of the last octet, seven zero slots behind it
            wait_free(oct);
            const unsigned ab = half_base(oct);
            S = chain_step(S, u0, std::false_type{}, proj_ok, ab, std::integral_constant<int, 0>{});
            for (int t = 1; t < 8; ++t) {
                asm volatile("ds_write_b32 %0, %1\n\tds_write_b32 %0, %1 offset:%2\n\tds_write_b32 %0, %1 offset:%3"
                             : : "v"(ab + t * RING_SLOT), "v"(0.f), "n"(RING_VAL), "n"(2 * RING_VAL) : "memory");
            }
            publish();
        }
        // ---------------- the chain wave's part of the slab: f | psi0bar | A ----------------
        const float sumS = sum64(accS);
        const float sumA = av == 0 ? sum64(accA) : 0.f;        // (z = e x / A belongs to the clip: counted with its first column)
        const float ftot = swapadd(facc, facc);               // half 0: f(h=0) + f(h=1)
        slab[4 * DD + (hb ? 2 * DPW : DPW) + i] = g;          // cotangent of psi_0: re in [DPW, 2DPW), im in [2DPW, 3DPW)
        if (!hb) slab[4 * DD + i] = ftot;
        if (lane == 0) {
            slab[4 * DD + 3 * DPW] = -(sumA / (A * A)) - sumS / A;        // (k_finalize removes the Q part of sumS: Dev::abar_fix)
            slab[4 * DD + 3 * DPW + 1] = 0.f;
        }
        return;
    }

    // ====================================================================== gradient wave
    __builtin_amdgcn_s_setprio(0);      // (equal or swapped priorities: no gain, profiles/r5_c3_ab_two_wave.log)
    typedef _Float16 hf8 __attribute__((ext_vector_type(8)));
    typedef _Float16 hf2 __attribute__((ext_vector_type(2)));
    v16f Rre = {}, Rim = {}, Qre = {}, Qim = {};
    constexpr float SB16 = 8192.f;                  // the unit vectors yhat, u: |.| <= 1
    float sR = 1.f, sQ = 1.f;                       // current scales of (a1 | a2) and of ybar (wave-uniform powers of two)
    auto pow2_of = [](float bound) {                // the largest power of two S with bound S < 2^12 (exponent clamped)
        int se = 12 - ((int)((__float_as_uint(bound) >> 23) & 0xFFu) - 126);
        se = se > 60 ? 60 : se < -60 ? -60 : se;
        return __uint_as_float((unsigned)(127 + se) << 23);
    };
    auto pow2_inv = [](float s2) { return __uint_as_float(0x7F000000u - __float_as_uint(s2)); };
    const unsigned aPart = lds_addr(&ring[w][0]) + (lane ^ 32) * 4;   // the partner component's lane
    const unsigned aSct0 = lds_addr(&sct[w][0][0]);
    for (int o = 0; o < n_oct; ++o) {
        {
            int spin = 0;
            while (flag_load2(aProd) < o + 1 && ++spin < SPIN_MAX) __builtin_amdgcn_s_sleep(2);
        }
        const unsigned ab = aRing0 + (unsigned)(o & 1) * RING_HALF, ap = aPart + (unsigned)(o & 1) * RING_HALF;
        // the octet's steps: slot t <-> step ktop - t (slots outside [0, N - 1] hold zero operands: any finite scalar will do)
        const int ktop = (u_top && o == 0) ? (N - 1) + (8 - u_top) : (o == n_oct - 1) ? 0 : (N - 1 - u_top) - 8 * (o - (u_top ? 1 : 0));
        int kl = ktop - (lane & 7);
        kl = kl < 0 ? 0 : (kl > N - 1 ? N - 1 : kl);
        const unsigned aSct = aSct0 + (unsigned)(((kl >> 6) & 1) * CH + (kl & (CH - 1))) * 8;
        float yb[8], yh[8], yhq[8], uu[8], uq[8];
        v2f stv;
#pragma unroll
        for (int t = 0; t < 8; t += 2) {             // two slots per instruction (ds_read2st64_b32: offsets in units of 64 dwords = one slot)
            asm volatile("ds_read2st64_b32 %0, %5 offset0:%7 offset1:%8\n\tds_read2st64_b32 %1, %5 offset0:%9 offset1:%10\n\t"
                         "ds_read2st64_b32 %2, %6 offset0:%9 offset1:%10\n\tds_read2st64_b32 %3, %5 offset0:%11 offset1:%12\n\t"
                         "ds_read2st64_b32 %4, %6 offset0:%11 offset1:%12"
                         : "=&v"(*reinterpret_cast<v2f*>(&yb[t])), "=&v"(*reinterpret_cast<v2f*>(&yh[t])), "=&v"(*reinterpret_cast<v2f*>(&yhq[t])),
                           "=&v"(*reinterpret_cast<v2f*>(&uu[t])), "=&v"(*reinterpret_cast<v2f*>(&uq[t]))
                         : "v"(ab), "v"(ap), "n"(t), "n"(t + 1), "n"(8 + t), "n"(8 + t + 1), "n"(16 + t), "n"(16 + t + 1) : "memory");   // units of 256 B
        }
        asm volatile("ds_read_b64 %0, %1" : "=&v"(stv) : "v"(aSct) : "memory");
        // every loaded register passes through a statement that HOLDS the wait (only the first one stalls): registers pinned by separate empty
        // statements may be copied by the compiler in front of the wait that makes them valid -- it happened in k_bwd_wave3's first form
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yb[0]), "+v"(yb[1]), "+v"(yb[2]), "+v"(yb[3]), "+v"(yb[4]), "+v"(yb[5]), "+v"(yb[6]), "+v"(yb[7]) :: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yh[0]), "+v"(yh[1]), "+v"(yh[2]), "+v"(yh[3]), "+v"(yh[4]), "+v"(yh[5]), "+v"(yh[6]), "+v"(yh[7]) :: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(yhq[0]), "+v"(yhq[1]), "+v"(yhq[2]), "+v"(yhq[3]), "+v"(yhq[4]), "+v"(yhq[5]), "+v"(yhq[6]), "+v"(yhq[7]) :: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(uu[0]), "+v"(uu[1]), "+v"(uu[2]), "+v"(uu[3]), "+v"(uu[4]), "+v"(uu[5]), "+v"(uu[6]), "+v"(uu[7]) :: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(uq[0]), "+v"(uq[1]), "+v"(uq[2]), "+v"(uq[3]), "+v"(uq[4]), "+v"(uq[5]), "+v"(uq[6]), "+v"(uq[7]), "+v"(stv) :: "memory");
        flag_store2(aConsL, o + 1);            // every ring read has landed: the half is free
        // a1 = ten yhat, a2 = s ybar (the scalars of slot t sit in lane t of the octet's scalar rows)
        float a1[8], a2[8], mQ = 0.f, mR = 0.f;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float sk = rdlane(stv.x, t), tn = rdlane(stv.y, t);
            a1[t] = tn * yh[t];
            a2[t] = sk * yb[t];
            mQ = fmaxf(mQ, fabsf(yb[t]));
            mR = fmaxf(mR, fmaxf(fabsf(a1[t]), fabsf(a2[t])));
        }
        mQ = wave_max(mQ);
        mR = wave_max(mR);
        // scales from the measured maxima: moved only when max * scale leaves [2^5, 2^15) (the accumulators follow by the exact ratio)
        {
            const float xq = mQ * sQ, xr = mR * sR;
            const bool chQ = mQ > 0.f && !(xq >= 32.f && xq < 32768.f), chR = mR > 0.f && !(xr >= 32.f && xr < 32768.f);
            if (chQ || chR) {
                const float nQ = chQ ? pow2_of(mQ) : sQ, nR = chR ? pow2_of(mR) : sR;
                const float fQ = nQ * pow2_inv(sQ), fR = nR * pow2_inv(sR);
#pragma unroll
                for (int r = 0; r < 16; ++r) { Rre[r] *= fR; Rim[r] *= fR; Qre[r] *= fQ; Qim[r] *= fQ; }
                sQ = nQ; sR = nR;
            }
        }
        // fp16 pieces: value v, register r holds slots 2 r (low half) and 2 r + 1 (high half)
        unsigned fH[7][4], fL[7][4];
        // p sc -> two fp16 pieces: one packed multiply, the hi pieces by conversion, each residual as ONE v_fma_mix_f32 (p sc - hi, the fp16
        // half read in place; the asm results feed a compiler-visible conversion, never an MFMA directly: DESIGN 4.3e)
        auto split_pair = [&](int v, int reg, float ve, float vo, float sc2) {
            const v2f t = mk2(ve, vo) * mk2(sc2, sc2);
            const hf2 hh = {(_Float16)t.x, (_Float16)t.y};
            const unsigned H = __builtin_bit_cast(unsigned, hh);
            float re, ro;
            asm("v_fma_mix_f32 %0, %2, %4, -%5 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
                "v_fma_mix_f32 %1, %3, %4, -%5 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
                : "=&v"(re), "=&v"(ro) : "v"(ve), "v"(vo), "v"(sc2), "v"(H));
            const hf2 ll = {(_Float16)re, (_Float16)ro};
            fH[v][reg] = H;
            fL[v][reg] = __builtin_bit_cast(unsigned, ll);
        };
        const float sgB = hb ? -SB16 : SB16;         // osig = the partner's value, negated in the upper half: the sign rides in the scale
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            split_pair(0, r, a1[2 * r], a1[2 * r + 1], sR);
            split_pair(1, r, yb[2 * r], yb[2 * r + 1], sQ);
            split_pair(2, r, a2[2 * r], a2[2 * r + 1], sR);
            split_pair(3, r, yh[2 * r], yh[2 * r + 1], SB16);
            split_pair(4, r, yhq[2 * r], yhq[2 * r + 1], sgB);
            split_pair(5, r, uu[2 * r], uu[2 * r + 1], SB16);
            split_pair(6, r, uq[2 * r], uq[2 * r + 1], sgB);
        }
        auto frag = [&](const unsigned (&f)[4]) { return __builtin_bit_cast(hf8, v4u{f[0], f[1], f[2], f[3]}); };
        auto mf3 = [&](v16f& acc, int ia, int ib) {  // hi hi' + hi lo' + lo hi'
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(fH[ia]), frag(fH[ib]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(fH[ia]), frag(fL[ib]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag(fL[ia]), frag(fH[ib]), acc, 0, 0, 0);
        };
        //   Rbar += 2 ebar y y^dagger + s ybar u^dagger ;  Qbar += ybar u^dagger
        //   Re(a b^dagger): A = a (split), B = b (split);  Im(a b^dagger): A = a (split), B = -b_osig  (sign applied once at the end)
        mf3(Rre, 0, 3);
        mf3(Rim, 0, 4);
        mf3(Qre, 1, 5);
        mf3(Qim, 1, 6);
        mf3(Rre, 2, 5);
        mf3(Rim, 2, 6);
    }
    // ---------------- the gradient wave's part of the slab: Rbar, Qbar ----------------
    const float uR = pow2_inv(sR) * (1.0f / SB16), uQ = pow2_inv(sQ) * (1.0f / SB16);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;   // C/D layout of the 32x32 MFMA: column = lane & 31
        const int oo = row * DPW + i;
        slab[oo] = Rre[r] * uR;
        slab[DD + oo] = -Rim[r] * uR;
        slab[2 * DD + oo] = Qre[r] * uQ;
        slab[3 * DD + oo] = -Qim[r] * uQ;
    }
}

hipError_t launch_bwd_wave2w(const Dev& P, const float* audio, hipStream_t s) {
    const unsigned nb = (unsigned)((P.B + WAVES - 1) / WAVES);
    hipLaunchKernelGGL(k_bwd_wave2w, dim3(nb), dim3(128 * WAVES), 0, s, P, audio);
    return hipGetLastError();
}

}  // namespace cmps
