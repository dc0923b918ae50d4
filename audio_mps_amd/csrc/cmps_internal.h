// Internal declarations shared by the HIP translation units of libcmps.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace cmps {

// ---------------------------------------------------------------------------------------------
// Per-kernel durations (cmps_set_option(CMPS_OPT_KERNEL_EVENTS, 1); cmps_kernel_times): while the option is on, every kernel the
// scan entry points launch is bracketed by two HIP events on the caller's stream.  Off (the default) nothing is recorded and a
// KScope is two pointer tests.  The C ABI functions set the thread-local pointer for the duration of a call.
// ---------------------------------------------------------------------------------------------
struct KTimer {
    struct Rec { const char* name; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<hipEvent_t> pool;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
};
extern thread_local KTimer* g_ktimer;
struct KScope {
    KTimer* t; hipStream_t s; const char* name; hipEvent_t a;
    KScope(const char* name_, hipStream_t s_) : t(g_ktimer), s(s_), name(name_), a(nullptr) {
        if (t) { a = t->get(); (void)hipEventRecord(a, s); }
    }
    ~KScope() {
        if (t) { hipEvent_t b = t->get(); (void)hipEventRecord(b, s); t->recs.push_back({name, a, b}); }
    }
};


// ---------------------------------------------------------------------------------------------
// Workspace layout (all offsets in bytes from the workspace base, every section 256-B aligned).
//
//   mats     : R [D][D] float2 | RT [D][D] float2 (RT[j][i] = R[i][j]) | Q [D][D] float2
//              (Q = -(dt sigma^2 / 2) R^dagger R, Hermitian) | psi0 [D] float2 | freqs [D] float | qflag (|Q|_F <= 2^-19)
//   ttab     : t_k, k = 0..N          float32, sequential sum (model.py:16,266,281)
//   dtk      : t_k - t_{k+1}          float32 (exact), k = 0..N-1
//   rho      : [N+1][DP] float2       rho_k[d] = exp(i (fl(f_d t_k) - fl(f_d t_{k+1}))), drift-corrected (cmps_prep.hip)
//   rfix     : [2][NC][DP] double2    scratch of the drift correction
//   stash    : [B][N][DP] float2      un-normalised rotating-frame state y_k (TRAIN only; block variant)
//   hst      : [B][N][64][2] float    wave variants' stash (same region as `stash`), 512 B per step: pairs (y_k, (R + R^dagger) y_k)
//                                     per real component n = 2 i + {re, im} (32-row layout) or per lane (16-row layout)
//   scal     : [B][NC][2][64] float   per 64-step chunk: |y_k|^2 and e_k, one step per lane (wave variant)
//   gops     : [pairs][N][4 DP] float   D > 32, TRAIN: ybar_k of the reverse scan for the gradient GEMM, which builds its
//                                     operands (te y | s ybar | ybar | y | u) itself: pair kernels [clip][re | im][DP], wide kernels
//                                     [wave][lane] (round 2 kept the five operands here in bf16, 2.5 x the bytes)
//   slabs    : [B][slab] float        per-clip gradient partials (TRAIN only)
//   sums     : [slab] float           reduced partials, followed by [32][slab] double first-pass partial sums
//   status   : [2] unsigned           flag words of the last cmps_psi_loss_bwd and their OR since the last cmps_psi_grad_status
//                                     (bit 0: a gradient sum is Inf / NaN, bit 1: the loss sum is) -- written by k_reduce_slabs / k_finalize
// DP = D rounded up to a multiple of 32 (components >= D are zero padding and stay exactly zero).
// ---------------------------------------------------------------------------------------------
struct Layout {
    int D, DP, B, T, N, flags;
    size_t off_R, off_RT, off_Q, off_QT, off_psi0, off_freqs, off_qflag, off_ttab, off_dtk, off_rho, off_rfix,
        off_stash, off_hst, off_scal, off_slabs, off_sums, off_gops, off_opmax, off_status, total;
    size_t slab_floats;  // 4*DP*DP + 3*DP + 2
};

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

inline int padded_D(int D) { return (D + 31) / 32 * 32; }   // zero padding to the next multiple of 32 (32, 64, 96, 128)

inline Layout make_layout(int D, int B, int T, int flags) {
    Layout L{};
    L.D = D; L.B = B; L.T = T; L.N = T - 1; L.flags = flags;
    // one layout serves both variants: tables are sized for the larger stride
    L.DP = padded_D(D);
    const size_t DP = (size_t)L.DP, N = (size_t)L.N;
    size_t o = 0;
    L.off_R = o;     o = align256(o + DP * DP * sizeof(float2));
    L.off_RT = o;    o = align256(o + DP * DP * sizeof(float2));
    L.off_Q = o;     o = align256(o + DP * DP * sizeof(float2));
    L.off_QT = o;    o = align256(o + DP * DP * sizeof(float2));
    L.off_psi0 = o;  o = align256(o + DP * sizeof(float2));
    L.off_freqs = o; o = align256(o + DP * sizeof(float));
    L.off_qflag = o; o = align256(o + sizeof(unsigned));
    L.off_ttab = o;  o = align256(o + (N + 1) * sizeof(float));
    L.off_dtk = o;   o = align256(o + (N + 64) * sizeof(float));
    L.off_rho = o;   o = align256(o + (N + 1) * DP * sizeof(float2));
    L.off_rfix = o;  o = align256(o + ((N + 63) / 64) * DP * 2 * sizeof(double2));
    L.slab_floats = 4 * DP * DP + 3 * DP + 2;
    L.off_stash = o;
    L.off_hst = o;
    L.off_scal = o;
    L.off_slabs = o;
    L.off_sums = o;
    L.off_gops = o;
    L.off_opmax = o;
    L.off_status = o;
    if (flags & 1) {
        // the two variants never run on the same stash: their layouts share one region
        // D > 32: the pair kernels (D = 128) keep (y, H y) per step, 16 B per component; the block kernels use half of it
        size_t stash_bytes = D > 32 ? (size_t)((B + 1) / 2 * 2) * N * DP * sizeof(float2) * 2   // whole pairs
                                    : (size_t)B * N * DP * sizeof(float2);
        if (D <= 32 && stash_bytes < (size_t)B * N * 128 * sizeof(float)) stash_bytes = (size_t)B * N * 128 * sizeof(float);
        L.off_hst = L.off_stash;
        o = align256(o + stash_bytes);
        L.off_scal = o;  o = align256(o + (size_t)B * ((N + 63) / 64) * 128 * sizeof(float));
        L.off_slabs = o; o = align256(o + (size_t)B * L.slab_floats * sizeof(float));
        L.off_sums = o;  o = align256(o + (L.slab_floats + 64) * sizeof(float) + 32 * L.slab_floats * sizeof(double));
        L.off_gops = o;
        // ybar rows: exactly [pairs][N][4 DP] floats (k_grad_gemm clamps every row it requests to the pair's range and its buffer
        // descriptors end at the last row; round 3's 32 rows of slack behind the section had no reader left -- ADVICE r3)
        if (D > 32) o = align256(o + (size_t)((B + 1) / 2) * N * 4 * DP * sizeof(float));
        // max |ybar| per pair (wide reverse scan -> the gradient GEMM's fp16 operand scale, CMPS_RANK1_F16X2)
        L.off_opmax = o;
        if (D > 32) o = align256(o + (size_t)((B + 1) / 2) * sizeof(float));
        L.off_status = o; o = align256(o + 2 * sizeof(unsigned));
    }
    L.total = o;
    return L;
}

// Device-side view handed to the kernels by value.
struct Dev {
    int D, DP, B, T, N;
    const float2* R;     // [DP][DP] row-major
    const float2* RT;    // [DP][DP], RT[j][i] = R[i][j]
    const float2* Q;     // [DP][DP] row-major, Hermitian (legacy mode: general complex)
    const float2* QT;    // [DP][DP] transpose of Q (legacy mode only)
    const float2* psi0;  // [DP]
    const float* freqs;  // [DP]
    const unsigned* qflag;  // [1]: 1 when |Q|_F <= 2^-19 (k_qflag, once per cmps_set_params): which instance of the chain16 kernels runs
    const float* ttab;   // [N+1]
    const float* dtk;    // [N]
    const float2* rho;   // [N][DP]
    float2* stash;       // [B][N][DP]
    float* hst;          // [B][N][64][2] (wave variant)
    int stash_layout;    // 0: stash [B][N][DP] float2 (block variant)  1: hst rows in lane order (cmps_wave16.hip)
                         // 2: pair rows (cmps_pair.hip)  3: hst rows of (y[n], (H y)[n]) pairs, n = 2 i + {re, im} (cmps_wave2.hip)
                         // 4: wide rows (cmps_wide.hip): [pair][step][y | H y][wave][lane] float
    float* scal;         // [B][NC][2][64]
    void* gops;          // ybar rows for the gradient GEMM (see the layout comment), D > 32 only
    float* opmax;        // [pairs] max |ybar| over the pair's steps and rows (written by k_bwd_wide), D > 32 only
    float* slabs;        // [B][slab]
    float* sums;         // [slab]
    size_t slab_floats;
    float A;
    const float* Adev;   // non-null: A lives in device memory (cmps_set_params_dev: device-resident optimiser step) and `A` is unset
    float dt;            // (float)delta_t (model.py:16; also the python-float factor of model.py:286)
    float c_half;        // (float)(-delta_t * sigma^2) / 2   (model.py:312)
    // RhoCMPS on the wide kernels (32 < D <= 128; cmps_wide.hip, round 5): the columns of every clip's rho are handed to the pure-state
    // kernels as VIRTUAL clips (virtual clip v = clip * phi_rank + column); then the initial vector of virtual clip v is phi0[v % phi_rank]
    // instead of psi0, dA's per-clip sum is taken once per real clip, and the reverse scan leaves every column's cotangent in gphi
    const float2* phi0;  // [phi_rank][DP] (zero rows pad an odd rank), or null: the pure-state model
    int phi_rank;        // columns per clip, even
    float* gphi;         // [virtual clips][re | im][DP]
    unsigned* status;    // [2]: flag word of the last reverse pass | OR since the last cmps_psi_grad_status (TRAIN workspaces; else null)
    int f16_shift;       // CMPS_OPT_F16_SCALE_SHIFT (diagnostic, 0): added to the exponent of the wave reverse scan's data-dependent fp16 scales
    int abar_fix;        // 1: the slabs' Abar holds -(sum_k Re(u^dagger (Q + s R^dagger) ybar)) / A (k_bwd_wave's merged mat-vec);
                         //    k_finalize adds Re sum_ij Q_ij Qbar_ji / A from the reduced Qbar
};

// ---------------------------------------------------------------------------------------------
// RhoCMPS (cmps_rho.hip): its own caller-provided workspace next to the main one.
//   phi0  : [rank][DP] float2         columns of rho_0 = sum_a phi_a phi_a^dagger (trace 1)
//   stash : [B][N][rank][DP] float2   un-normalised rotating-frame columns y_a of every step (TRAIN / sampling states)
//   slabs : [B][slab] float           pure-state slab followed by the 2 rank DP cotangents of the initial columns
//   sums  : [slab] float + [32][slab] double first-pass partials
// ---------------------------------------------------------------------------------------------
struct RhoLayout {
    int rank;
    size_t off_phi0, off_stash, off_scal, off_slabs, off_sums, off_p1, off_cols, total, slab_floats;
    // the wide (virtual-clip) path of 32 < D <= 128, TRAIN workspaces: vrank = rank rounded up to even (0: not available)
    size_t off_wslabs, off_wsums;       // D <= 32, TRAIN: one pure-state slab per (clip, column) for the wave reverse scan on virtual clips
    int vrank;
    size_t off_vphi, off_vstash, off_vgops, off_vopmax, off_vscal, off_rscal, off_vslabs, off_vsums, off_vaudio, off_gphi, vslab_floats;
};

// RhoCMPS on the wide kernels: every pair of columns of a clip is one broadcast vector in LDS, two buffers (cmps_wide.hip::k_fwd_wide_rho)
inline size_t rho_wide_lds(int D, int rank) {
    const size_t PD = (size_t)padded_D(D), vr = (size_t)((rank + 1) / 2 * 2);
    return 2 * (vr / 2) * (8 * (PD / 8 + 1)) * 16 + 2 * (PD / 16) * sizeof(float) + 64;
}
inline bool rho_wide_ok(int D, int rank, int flags) { return D > 32 && D <= 128 && (flags & 1) && rho_wide_lds(D, rank) <= 120 * 1024; }

// rank * D above which the block kernels' column arrays (4 rank D complex numbers in the reverse scan) no longer fit into 160 KB of
// LDS and live in the workspace instead (RhoDev::cols): the reference's default rank = D (model.py:62-65) from D = 72 upwards
constexpr size_t RHO_LDS_COLS_MAX = 5000;
// LDS a workgroup of the block rho kernels may use for its column arrays, and the ONE predicate both the launchers (cmps_rho.hip:
// cols_if_needed) and the argument checks (cmps_capi.hip) use for "this kernel's `arrays` column arrays go to the workspace"
// (forward 2, reverse scan 4, sampler 3 arrays of rank * D complex numbers): the workspace section is provided from
// RHO_LDS_COLS_MAX upwards (sized for the reverse scan), but a kernel spills only when ITS request exceeds the LDS
constexpr size_t RHO_LDS_MAX = 160 * 1024;
inline bool rho_cols_spill(int arrays, int rank, int D) { return (size_t)arrays * rank * D * sizeof(float2) + 128 > RHO_LDS_MAX; }

inline RhoLayout make_rho_layout(int D, int rank, int B, int T, int flags) {
    RhoLayout L{};
    const size_t DP = (size_t)padded_D(D), N = (size_t)(T - 1), r = (size_t)rank;
    L.rank = rank;
    L.slab_floats = 4 * DP * DP + 3 * DP + 2 + 2 * r * DP;
    size_t o = 0;
    L.off_phi0 = o; o = align256(o + r * DP * sizeof(float2));
    L.off_stash = L.off_scal = L.off_slabs = L.off_sums = o;
    if (flags & 1) {
        // D <= 32: the wave kernels keep (y_a own, (H y_a) own) per lane, 512 B per column and step
        L.off_stash = o; o = align256(o + (size_t)B * N * r * DP * sizeof(float2) * (D <= 32 ? 2 : 1));
        L.off_scal = o;  o = align256(o + (size_t)B * ((N + 63) / 64) * 128 * sizeof(float));
        L.off_slabs = o; o = align256(o + (size_t)B * L.slab_floats * sizeof(float));
        L.off_sums = o;  o = align256(o + (L.slab_floats + 64) * sizeof(float) + 32 * L.slab_floats * sizeof(double));
        // D <= 32: sum_k 2 ebar_k Y^T Y (real 64 x 64 form, four C/D tiles per clip), accumulated by k_fwd_rho_mfma for the reverse scan
        L.off_p1 = o;    if (D <= 32) o = align256(o + (size_t)B * 4096 * sizeof(float));
        L.off_wslabs = L.off_wsums = o;
        if (D <= 32) {
            const size_t psl = 4 * DP * DP + 3 * DP + 2;
            L.off_wslabs = o; o = align256(o + (size_t)B * r * psl * sizeof(float));
            L.off_wsums = o;  o = align256(o + (psl + 64) * sizeof(float) + 32 * psl * sizeof(double));
        }
    }
    // column arrays of the block kernels when they do not fit into LDS: [B][4][rank][D] float2 (forward 2, sampler 3, reverse 4)
    L.off_cols = o;
    if (r * (size_t)D > RHO_LDS_COLS_MAX) o = align256(o + (size_t)B * 4 * r * D * sizeof(float2));
    L.vrank = 0;
    if (rho_wide_ok(D, rank, flags)) {
        const size_t vr = (size_t)((rank + 1) / 2 * 2), vB = (size_t)B * vr, vp = vB / 2, NC = (N + 63) / 64;
        L.vrank = (int)vr;
        L.vslab_floats = 4 * DP * DP + 3 * DP + 2;
        L.off_vphi = o;   o = align256(o + vr * DP * sizeof(float2));
        L.off_vstash = o; o = align256(o + vp * N * 8 * DP * sizeof(float));         // [virtual pair][step][y | H y][4 DP]
        L.off_vgops = o;  o = align256(o + vp * N * 4 * DP * sizeof(float));         // ybar rows
        L.off_vopmax = o; o = align256(o + vp * sizeof(float));
        L.off_vscal = o;  o = align256(o + vB * NC * 128 * sizeof(float));           // (|y|^2 of the clip, e of the clip) per virtual clip
        L.off_rscal = o;  o = align256(o + (size_t)B * NC * 128 * sizeof(float));    // the same per real clip
        L.off_vslabs = o; o = align256(o + vp * L.vslab_floats * sizeof(float));
        L.off_vsums = o;  o = align256(o + (L.vslab_floats + 64) * sizeof(float) + 32 * L.vslab_floats * sizeof(double));
        L.off_vaudio = o; o = align256(o + vB * (size_t)T * sizeof(float));
        L.off_gphi = o;   o = align256(o + vB * 2 * DP * sizeof(float));
    }
    L.total = o;
    return L;
}

struct RhoDev {
    int rank;
    int stash_layout;    // 0: [B][N][rank][DP] float2 (cmps_rho.hip)  1: [B][N][rank][64] (y own, H y own) (cmps_rho_wave.hip)
                         // 2: [B][N][rank][64] pairs (y[n], (H y)[n]), n = 2 i + {re, im} (cmps_rho_mfma.hip)
                         // 3: the wide kernels' rows, one vector per PAIR of columns: vstash [(b vrank + a) / 2][N][y | H y][4 DP] (cmps_wide.hip)
    float* wslabs; float* wsums;   // D <= 32: the wave reverse scan's slabs, one per (clip, column), and their reduction buffer
    int vrank;           // rank rounded up to even when the wide path's sections exist (else 0)
    float2* vphi;        // [vrank][DP]
    float* vstash; float* vgops; float* vopmax; float* vscal; float* rscal; float* vslabs; float* vsums; float* vaudio; float* gphi;
    size_t vslab_floats;
    float* scal;         // [B][NC][2][64]: tr rho'_k and e_k, one step per lane (wave kernels)
    float* p1;           // [B][4][16][64]: the forward's part of Rbar (cmps_rho_mfma.hip), raw C/D tiles
    const float2* phi0;  // [rank][DP]
    float2* cols;        // [cols_blocks][4][rank][D] column arrays of the block kernels when rank * D exceeds the LDS (else null)
    int cols_blocks;
    float2* stash;       // [B][N][rank][DP]
    float* slabs;        // [B][slab]
    float* sums;         // [slab]
    size_t slab_floats;
};

// ---- launchers (each returns the hipError_t of its launches) ----
hipError_t launch_pack_phi(const Dev& P, const RhoDev& W, const float* re, const float* im, hipStream_t s);
hipError_t launch_fwd_rho(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, hipStream_t s);
hipError_t launch_bwd_rho(const Dev& P, const RhoDev& W, const float* audio, hipStream_t s);
hipError_t launch_finalize_rho(const Dev& P, const RhoDev& W, float* grad_out, hipStream_t s);
hipError_t launch_states_rho(const Dev& P, const RhoDev& W, int B, int steps, float* rho_out, float* purity_out,
                             hipStream_t s);
hipError_t launch_update_ancilla_rho(const Dev& P, const float* rho_in, const float* signal, float t, int B,
                                     float* rho_out, hipStream_t s);
hipError_t launch_sample_rho(const Dev& P, const RhoDev& W, const float* noise, int n, int length, float* out,
                             bool save, hipStream_t s);
hipError_t launch_fwd_legacy_wave(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s);
hipError_t launch_bwd_legacy_wave(const Dev& P, const float* audio, int rank1_mode, hipStream_t s);
hipError_t launch_legacy_tables(const Dev& P, float2* psi0, float* dtk, float2* rho, hipStream_t s);
hipError_t launch_fwd_rho_wave(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, hipStream_t s);
hipError_t launch_fwd_rho_mfma(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool save, bool f16, bool grad1, hipStream_t s);
hipError_t launch_bwd_rho_mfma(const Dev& P, const RhoDev& W, const float* audio, hipStream_t s);
hipError_t launch_sample_rho_mfma(const Dev& P, const RhoDev& W, const float* noise, int n, int length, float* out, bool save,
                                  bool f16, hipStream_t s);
hipError_t launch_bwd_rho_wave(const Dev& P, const RhoDev& W, const float* audio, hipStream_t s);
hipError_t launch_prep(const Dev& P, const float* R_re, const float* R_im, const float* freqs,
                       const float* psi0_re, const float* psi0_im, float dt, bool rebuild_ttab,
                       float* ttab, float* dtk, float2* R, float2* RT, float2* Q, float2* psi0,
                       float* freqs_out, float2* rho, double2* rfix, hipStream_t s);

hipError_t launch_fwd_block(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s);
hipError_t launch_bwd_block(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_bwd_wave(const Dev& P, const float* audio, int rank1_mode, hipStream_t s);
hipError_t launch_bwd_wave2w(const Dev& P, const float* audio, hipStream_t s);    // cmps_wave_bwd2.hip: two waves per clip (F16X2 sums)
hipError_t launch_fwd_wave2(const Dev& P, const float* audio, float* loss, bool save, bool hf16, hipStream_t s);
hipError_t launch_fwd_wave16(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s);
hipError_t launch_bwd_wave16(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_fwd_pair(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s);
hipError_t launch_bwd_pair(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_grad_pair(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_fwd_wide(const Dev& P, const float* audio, float* loss, bool save, bool hy_f16, bool chain_mfma, hipStream_t s);
hipError_t launch_fwd_wide_legacy(const Dev& P, const float* audio, float* loss, bool save, bool hy_f16, hipStream_t s);
hipError_t launch_bwd_wide_legacy(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_grad_wide_legacy(const Dev& P, const float* audio, bool f16, hipStream_t s);
hipError_t launch_bwd_rho_virtual_wave(const Dev& P, const RhoDev& W, const float* audio, const float* loss, float* grad_out, int rank1_mode, int bwd_waves,
                                       hipStream_t s);
hipError_t launch_fwd_rho_wide(const Dev& P, const RhoDev& W, const float* audio, float* loss, bool hy_f16, hipStream_t s);
hipError_t launch_bwd_rho_wide(const Dev& P, const RhoDev& W, const float* loss, float* grad_out, int pieces, hipStream_t s);
hipError_t launch_fwd_chain16(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_bwd_chain16(const Dev& P, const float* audio, hipStream_t s);    // cmps_pair.hip: the wide family's chain on the matrix cores
hipError_t launch_bwd_wide(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_grad_wide(const Dev& P, const float* audio, int pieces, hipStream_t s);
hipError_t launch_finalize_only(const Dev& P, const float* loss, float* grad_out, hipStream_t s);
hipError_t launch_reduce_finalize(const Dev& P, const float* loss, float* grad_out, hipStream_t s);
hipError_t launch_update_ancilla(const Dev& P, const float* psi_in, const float* signal, float t,
                                 int B, float* psi_out, hipStream_t s);
hipError_t launch_states(const Dev& P, int B, float* psi_out, hipStream_t s);
hipError_t launch_reduce_only(const Dev& P, hipStream_t s);
hipError_t launch_pack_legacy(const Dev& P, const float* Rr, const float* Qre, const float* Qim, float2* R,
                              float2* RT, float2* Q, float2* QT, hipStream_t s);
hipError_t launch_fwd_legacy(const Dev& P, const float* audio, float* loss, bool save, hipStream_t s);
hipError_t launch_bwd_legacy(const Dev& P, const float* audio, hipStream_t s);
hipError_t launch_finalize_legacy(const Dev& P, const float* loss, float* grad_out, hipStream_t s);
hipError_t launch_sample_wave(const Dev& P, const float* noise, int n, int length, float* out, hipStream_t s);
hipError_t launch_sample_wide(const Dev& P, const float* noise, int n, int length, float* out, hipStream_t s);
hipError_t launch_sample_block(const Dev& P, const float* noise, int n, int length, float* out, hipStream_t s);

size_t apply_step_scratch_bytes(int D);
hipError_t launch_apply_step(int D, bool apply, double inv_batch, double lr_t, double beta1, double beta2, double eps, double h_reg,
                             double r_reg, double c_r, double c_h, bool with_reg, float* vars, float* am, float* av,
                             const float* grad_sums, float* params_out, float* losses_out, double* scratch, hipStream_t s);

// model.A (model.py:19) as the kernels see it: a launch argument, or -- device-resident training -- a word of device memory
__device__ __forceinline__ float dev_A(const Dev& P) { return P.Adev ? *P.Adev : P.A; }

// ---- small complex helpers (device) ----
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cmul_conj_a(float2 a, float2 b) {  // conj(a) * b
    return make_float2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ float2 cfma(float2 a, float2 b, float2 c) {  // a*b + c
    return make_float2(fmaf(a.x, b.x, fmaf(-a.y, b.y, c.x)), fmaf(a.x, b.y, fmaf(a.y, b.x, c.y)));
}
__device__ __forceinline__ float2 cfma_conj_a(float2 a, float2 b, float2 c) {  // conj(a)*b + c
    return make_float2(fmaf(a.x, b.x, fmaf(a.y, b.y, c.x)), fmaf(a.x, b.y, fmaf(-a.y, b.x, c.y)));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 cscale(float s, float2 a) { return make_float2(s * a.x, s * a.y); }

}  // namespace cmps
