// C ABI of libcmps.so (see include/cmps.h for the contract and the reference lines each entry replaces).
#include "../../include/cmps.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>

#include "cmps_internal.h"

using namespace cmps;

namespace cmps { thread_local KTimer* g_ktimer = nullptr; }

struct cmps_handle_s {
    int D = 0;
    int variant_req = CMPS_VARIANT_AUTO;
    int rank1_mode = CMPS_RANK1_DEFAULT;
    int wide_chain = CMPS_WIDE_CHAIN_MFMA;
    int f16_shift = 0;         // CMPS_OPT_F16_SCALE_SHIFT (diagnostic)
    int bwd_waves = 2;             // CMPS_OPT_BWD_WAVES
    bool rho_fwd_grad1 = false;    // the last GEMM forward accumulated RhoDev::p1
    bool rho_virtual_bwd = true;   // CMPS_OPT_RHO_BWD: the RhoCMPS GEMM forward's reverse sweep on virtual clips of k_bwd_wave (else k_bwd_rho_mfma)
    bool params_set = false;
    bool legacy = false;       // the tables currently hold the legacy AudioMPS arithmetic (cmps_legacy_set_params)
    bool fwd_saved = false;
    int saved_B = 0, saved_T = 0, saved_variant = 0;
    const float* saved_audio = nullptr;
    float* saved_loss = nullptr;
    Layout L{};
    Dev P{};
    char* ws = nullptr;
    // time-table cache key: the table is rebuilt only when (workspace, N, dt) changes
    char* tt_ws = nullptr;
    int tt_N = -1;
    float tt_dt = 0.f;
    // RhoCMPS state (cmps_rho_set_state)
    bool rho_set = false;
    bool rho_saved = false;       // the rho stash holds the columns of rho_saved_B x rho_saved_steps steps
    bool rho_bwd_ok = false;      // ... written by cmps_rho_loss_fwd (not by the sampler)
    int rho_B = 0, rho_T = 0, rho_flags = 0, rho_saved_B = 0, rho_saved_steps = 0;
    RhoLayout RL{};
    RhoDev W{};
    cmps::KTimer* ktimer = nullptr;   // non-null: CMPS_OPT_KERNEL_EVENTS is on
    std::string err;
};

namespace {

int fail(cmps_handle_t h, int code, const char* msg) {
    if (h) h->err = msg;
    return code;
}

void ktimer_free(cmps::KTimer* t) {
    for (auto& r : t->recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    for (auto e : t->pool) (void)hipEventDestroy(e);
    delete t;
}
// sets the thread-local timer for the duration of one C ABI call
struct KBind {
    explicit KBind(cmps_handle_t h) { cmps::g_ktimer = h ? h->ktimer : nullptr; }
    ~KBind() { cmps::g_ktimer = nullptr; }
};

int fail_hip(cmps_handle_t h, hipError_t e, const char* where) {
    if (h) {
        h->err = std::string(where) + ": " + hipGetErrorString(e);
    }
    return CMPS_ERR_HIP;
}

int resolve_variant(const cmps_handle_s* h) {
    if (h->variant_req == CMPS_VARIANT_BLOCK) return CMPS_VARIANT_BLOCK;
    if (h->variant_req == CMPS_VARIANT_WAVE) return h->D <= 32 ? CMPS_VARIANT_WAVE : CMPS_VARIANT_BLOCK;
    if (h->variant_req == CMPS_VARIANT_WAVE32) return h->D <= 32 ? CMPS_VARIANT_WAVE32 : CMPS_VARIANT_BLOCK;
    if (h->variant_req == CMPS_VARIANT_PAIR) return h->D > 32 ? CMPS_VARIANT_PAIR : CMPS_VARIANT_BLOCK;
    if (h->variant_req == CMPS_VARIANT_WIDE) return h->D > 32 ? CMPS_VARIANT_WIDE : CMPS_VARIANT_BLOCK;
    // AUTO: the reference's float32 arithmetic at every D (fp32-faithful products, see cmps.h); the bf16-operand MFMA kernels
    // of 32 < D <= 128 are opt-in (they change the arithmetic)
    return h->D <= 32 ? CMPS_VARIANT_WAVE : CMPS_VARIANT_WIDE;
}

// CMPS_OPT_RANK1 as the wave-per-clip reverse scans understand it: exact fp32, two bf16 pieces, two fp16 pieces (the 32-row kernel of
// the PsiCMPS arithmetic; the legacy mode maps it to three bf16 pieces), or (every other value) three bf16 pieces
int wave_rank1(int mode) {
    if (mode == CMPS_RANK1_DEFAULT) return CMPS_RANK1_F16X2;
    return mode == CMPS_RANK1_EXACT_F32 || mode == CMPS_RANK1_BF16X2 || mode == CMPS_RANK1_F16X2 ? mode : CMPS_RANK1_BF16X3;
}

}  // namespace

extern "C" {

int cmps_version(void) { return 300; }

int cmps_create(int D, cmps_handle_t* out) {
    if (!out) return CMPS_ERR_BAD_ARG;
    *out = nullptr;
    if (D < 1 || D > 128) return CMPS_ERR_UNSUPPORTED_D;
    cmps_handle_s* h = new (std::nothrow) cmps_handle_s();
    if (!h) return CMPS_ERR_BAD_ARG;
    h->D = D;
    *out = h;
    return CMPS_OK;
}

int cmps_destroy(cmps_handle_t h) {
    if (h && h->ktimer) ktimer_free(h->ktimer);
    delete h;
    return CMPS_OK;
}

const char* cmps_last_error(cmps_handle_t h) { return h ? h->err.c_str() : "null handle"; }

int cmps_set_variant(cmps_handle_t h, int variant) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (variant < CMPS_VARIANT_AUTO || variant > CMPS_VARIANT_WIDE)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_variant: unknown variant");
    if ((variant == CMPS_VARIANT_PAIR || variant == CMPS_VARIANT_WIDE) && h->D <= 32)
        return fail(h, CMPS_ERR_UNSUPPORTED_D, "cmps_set_variant: the pair (bf16 MFMA) and wide (float32) variants are for 32 < D <= 128");
    if ((variant == CMPS_VARIANT_WAVE || variant == CMPS_VARIANT_WAVE32) && h->D > 32)
        return fail(h, CMPS_ERR_UNSUPPORTED_D, "cmps_set_variant: the wave variant needs D <= 32");
    h->variant_req = variant;
    return CMPS_OK;
}

int cmps_get_variant(cmps_handle_t h) { return h ? resolve_variant(h) : 0; }

int cmps_set_option(cmps_handle_t h, int option, int value) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (option == CMPS_OPT_RANK1) {
        if (value < CMPS_RANK1_EXACT_F32 || value > CMPS_RANK1_DEFAULT)
            return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_option: unknown value for CMPS_OPT_RANK1");
        h->rank1_mode = value;
        return CMPS_OK;
    }
    if (option == CMPS_OPT_WIDE_CHAIN) {
        if (value < CMPS_WIDE_CHAIN_VALU || value > CMPS_WIDE_CHAIN_MFMA_FWD)
            return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_option: unknown value for CMPS_OPT_WIDE_CHAIN");
        h->wide_chain = value;
        return CMPS_OK;
    }
    if (option == CMPS_OPT_BWD_WAVES) {
        if (value != 1 && value != 2) return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_option: CMPS_OPT_BWD_WAVES takes 1 or 2");
        h->bwd_waves = value;
        return CMPS_OK;
    }
    if (option == CMPS_OPT_RHO_BWD) {
        if (value != CMPS_RHO_BWD_VIRTUAL && value != CMPS_RHO_BWD_GEMM) return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_option: unknown value for CMPS_OPT_RHO_BWD");
        h->rho_virtual_bwd = value == CMPS_RHO_BWD_VIRTUAL;
        return CMPS_OK;
    }
    if (option == CMPS_OPT_F16_SCALE_SHIFT) {
        if (value < -40 || value > 40) return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_option: CMPS_OPT_F16_SCALE_SHIFT takes -40 .. 40");
        h->f16_shift = value;
        return CMPS_OK;
    }
    if (option == CMPS_OPT_KERNEL_EVENTS) {
        if (value != 0 && value != 1) return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_option: CMPS_OPT_KERNEL_EVENTS takes 0 or 1");
        if (value && !h->ktimer) h->ktimer = new cmps::KTimer();
        if (!value && h->ktimer) { ktimer_free(h->ktimer); h->ktimer = nullptr; }
        return CMPS_OK;
    }
    return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_option: unknown option");
}

int cmps_get_option(cmps_handle_t h, int option) {
    if (!h) return -1;
    if (option == CMPS_OPT_RANK1) return h->rank1_mode;
    if (option == CMPS_OPT_KERNEL_EVENTS) return h->ktimer ? 1 : 0;
    if (option == CMPS_OPT_WIDE_CHAIN) return h->wide_chain;
    if (option == CMPS_OPT_F16_SCALE_SHIFT) return h->f16_shift;
    if (option == CMPS_OPT_BWD_WAVES) return h->bwd_waves;
    if (option == CMPS_OPT_RHO_BWD) return h->rho_virtual_bwd ? CMPS_RHO_BWD_VIRTUAL : CMPS_RHO_BWD_GEMM;
    return -1;
}

size_t cmps_workspace_bytes(int D, int B, int T, int flags) {
    if (D < 1 || D > 128 || B < 1 || T < 2) return 0;
    return make_layout(D, B, T, flags & ~(CMPS_WS_FRESH | CMPS_WS_REUSE_TABLES)).total;
}

static int set_params_impl(cmps_handle_t h, const float* R_re_dev, const float* R_im_dev,
                           const float* freqs_dev, const float* psi0_re_dev, const float* psi0_im_dev,
                           float A, const float* A_dev, double sigma, double delta_t, int T, int B_max, int flags,
                           void* workspace_dev, size_t workspace_bytes, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!R_re_dev || !R_im_dev || !freqs_dev || !psi0_re_dev || !psi0_im_dev)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_params: null parameter pointer");
    if (T < 2 || B_max < 1) return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_params: need T >= 2 and B_max >= 1");
    if (!workspace_dev) return fail(h, CMPS_ERR_WORKSPACE, "cmps_set_params: null workspace");
    // the cached time table is trusted only when the caller says the workspace is untouched (and never with CMPS_WS_FRESH)
    const bool fresh = (flags & CMPS_WS_FRESH) != 0 || (flags & CMPS_WS_REUSE_TABLES) == 0;
    flags &= ~(CMPS_WS_FRESH | CMPS_WS_REUSE_TABLES);
    Layout L = make_layout(h->D, B_max, T, flags);
    if (workspace_bytes < L.total) {
        char buf[160];
        snprintf(buf, sizeof buf, "cmps_set_params: workspace has %zu bytes, needs %zu", workspace_bytes, L.total);
        return fail(h, CMPS_ERR_WORKSPACE, buf);
    }
    if (((uintptr_t)workspace_dev & 255) != 0)
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_set_params: workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace_dev);
    Dev P{};
    P.D = L.D; P.DP = L.DP; P.B = B_max; P.T = T; P.N = L.N;
    P.R = reinterpret_cast<float2*>(ws + L.off_R);
    P.RT = reinterpret_cast<float2*>(ws + L.off_RT);
    P.Q = reinterpret_cast<float2*>(ws + L.off_Q);
    P.QT = reinterpret_cast<float2*>(ws + L.off_QT);
    P.psi0 = reinterpret_cast<float2*>(ws + L.off_psi0);
    P.freqs = reinterpret_cast<float*>(ws + L.off_freqs);
    P.qflag = reinterpret_cast<unsigned*>(ws + L.off_qflag);
    P.ttab = reinterpret_cast<float*>(ws + L.off_ttab);
    P.dtk = reinterpret_cast<float*>(ws + L.off_dtk);
    P.rho = reinterpret_cast<float2*>(ws + L.off_rho);
    P.stash = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float2*>(ws + L.off_stash) : nullptr;
    P.hst = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_hst) : nullptr;
    P.scal = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_scal) : nullptr;
    P.gops = ((flags & CMPS_WS_TRAIN) && L.D > 32) ? static_cast<void*>(ws + L.off_gops) : nullptr;
    P.opmax = ((flags & CMPS_WS_TRAIN) && L.D > 32) ? reinterpret_cast<float*>(ws + L.off_opmax) : nullptr;
    P.slabs = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_slabs) : nullptr;
    P.sums = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_sums) : nullptr;
    P.status = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<unsigned*>(ws + L.off_status) : nullptr;
    P.slab_floats = L.slab_floats;
    P.A = A;
    P.Adev = A_dev;
    // model.py:312: `- self.delta_t * self.sigma**2` is a Python float (double), cast to complex64,
    // multiplied in, then divided by 2. (exact halving)
    P.c_half = (float)(-delta_t * sigma * sigma) / 2.0f;
    const float dt = (float)delta_t;  // model.py:16
    P.dt = dt;
    const bool rebuild = fresh || !(h->tt_ws == ws && h->tt_N == L.N && h->tt_dt == dt);
    hipError_t e = launch_prep(P, R_re_dev, R_im_dev, freqs_dev, psi0_re_dev, psi0_im_dev, dt, rebuild,
                               const_cast<float*>(P.ttab), const_cast<float*>(P.dtk),
                               const_cast<float2*>(P.R), const_cast<float2*>(P.RT),
                               const_cast<float2*>(P.Q), const_cast<float2*>(P.psi0),
                               const_cast<float*>(P.freqs), const_cast<float2*>(P.rho),
                               reinterpret_cast<double2*>(ws + L.off_rfix), static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_set_params");
    if (P.status && (fresh || h->ws != ws)) {        // a workspace this handle cannot vouch for: the flag words start at zero
        e = hipMemsetAsync(P.status, 0, 2 * sizeof(unsigned), static_cast<hipStream_t>(stream));
        if (e != hipSuccess) return fail_hip(h, e, "cmps_set_params (status words)");
    }
    h->tt_ws = ws; h->tt_N = L.N; h->tt_dt = dt;
    h->L = L; h->P = P; h->ws = ws;
    h->params_set = true;
    h->legacy = false;
    h->fwd_saved = false;
    h->rho_set = false;               // the columns of rho_0 are handed over again after every parameter change
    h->rho_saved = false;
    return CMPS_OK;
}

int cmps_set_params(cmps_handle_t h, const float* R_re_dev, const float* R_im_dev,
                    const float* freqs_dev, const float* psi0_re_dev, const float* psi0_im_dev,
                    float A, double sigma, double delta_t, int T, int B_max, int flags,
                    void* workspace_dev, size_t workspace_bytes, void* stream) {
    return set_params_impl(h, R_re_dev, R_im_dev, freqs_dev, psi0_re_dev, psi0_im_dev, A, nullptr, sigma, delta_t, T, B_max, flags,
                           workspace_dev, workspace_bytes, stream);
}

int cmps_set_params_dev(cmps_handle_t h, const float* params_dev, double sigma, double delta_t, int T, int B_max, int flags,
                        void* workspace_dev, size_t workspace_bytes, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!params_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_set_params_dev: null parameter buffer");
    const size_t DD = (size_t)h->D * h->D, D = (size_t)h->D;
    return set_params_impl(h, params_dev, params_dev + DD, params_dev + 2 * DD, params_dev + 2 * DD + D, params_dev + 2 * DD + 2 * D,
                           0.f, params_dev + 2 * DD + 3 * D, sigma, delta_t, T, B_max, flags, workspace_dev, workspace_bytes, stream);
}

size_t cmps_apply_step_scratch_bytes(int D) { return (D < 1 || D > 128) ? 0 : apply_step_scratch_bytes(D); }

int cmps_psi_apply_step(cmps_handle_t h, float* vars_dev, float* adam_m_dev, float* adam_v_dev, const float* grad_sums_dev,
                        double global_batch, double lr_t, double beta1, double beta2, double epsilon, double h_reg, double r_reg,
                        double c_r, double c_h, int with_reg, float* params_dev, float* losses_dev, void* scratch_dev,
                        void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!vars_dev || !params_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_apply_step: null variable / parameter buffer");
    const bool apply = grad_sums_dev != nullptr;
    if (apply && (!adam_m_dev || !adam_v_dev || !losses_dev || !scratch_dev || !(global_batch > 0.0)))
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_apply_step: an update needs the Adam slots, losses_dev, scratch_dev and a positive batch");
    if (((uintptr_t)scratch_dev & 7) != 0) return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_apply_step: scratch_dev must be 8-byte aligned");
    const hipError_t e = launch_apply_step(h->D, apply, apply ? 1.0 / global_batch : 0.0, lr_t, beta1, beta2, epsilon, h_reg, r_reg, c_r,
                                           c_h, with_reg != 0, vars_dev, adam_m_dev, adam_v_dev, grad_sums_dev, params_dev, losses_dev,
                                           static_cast<double*>(scratch_dev), static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_apply_step");
    return CMPS_OK;
}

int cmps_psi_loss_fwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* loss_dev,
                      int save_for_bwd, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || h->legacy) return fail(h, CMPS_ERR_STATE, "cmps_psi_loss_fwd: call cmps_set_params first");
    if (!audio_dev || !loss_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_loss_fwd: null pointer");
    if (T != h->L.T) return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_loss_fwd: T differs from cmps_set_params");
    if (B < 1 || B > h->L.B) return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_loss_fwd: B outside [1, B_max]");
    if (save_for_bwd && !(h->L.flags & CMPS_WS_TRAIN))
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_psi_loss_fwd: save_for_bwd needs a CMPS_WS_TRAIN workspace");
    Dev P = h->P;
    P.B = B;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int variant = resolve_variant(h);
    KBind kb(h);
    hipError_t e;
    if (variant == CMPS_VARIANT_WAVE && h->D <= 16) {
        KScope ks("k_fwd_wave16", s);
        e = launch_fwd_wave16(P, audio_dev, loss_dev, save_for_bwd != 0, s);
    } else if (variant == CMPS_VARIANT_WAVE || variant == CMPS_VARIANT_WAVE32) {
        KScope ks("k_fwd_wave2", s);
        // the loss product's pieces follow CMPS_OPT_RANK1 like the reverse scan's rank-1 sums: two fp16 pieces for F16X2 / DEFAULT
        e = launch_fwd_wave2(P, audio_dev, loss_dev, save_for_bwd != 0, wave_rank1(h->rank1_mode) == CMPS_RANK1_F16X2, s);
    } else if (variant == CMPS_VARIANT_PAIR) {
        KScope ks("k_fwd_pair", s);
        e = launch_fwd_pair(P, audio_dev, loss_dev, save_for_bwd != 0, s);
    } else if (variant == CMPS_VARIANT_WIDE) {
        // k_fwd_wide, k_hy_wide, k_loss_wide (scopes inside); the loss product's pieces follow CMPS_OPT_RANK1 like the gradient GEMM's
        e = launch_fwd_wide(P, audio_dev, loss_dev, save_for_bwd != 0,
                            h->rank1_mode == CMPS_RANK1_DEFAULT || h->rank1_mode == CMPS_RANK1_F16X2, h->wide_chain != CMPS_WIDE_CHAIN_VALU, s);
    } else {
        KScope ks("k_fwd_block", s);
        e = launch_fwd_block(P, audio_dev, loss_dev, save_for_bwd != 0, s);
    }
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_loss_fwd");
    h->fwd_saved = save_for_bwd != 0;
    h->saved_B = B; h->saved_T = T; h->saved_audio = audio_dev; h->saved_loss = loss_dev;
    h->saved_variant = variant;
    h->P.stash_layout = (variant == CMPS_VARIANT_WAVE && h->D <= 16) ? 1
                      : (variant == CMPS_VARIANT_WAVE || variant == CMPS_VARIANT_WAVE32) ? 3
                      : (variant == CMPS_VARIANT_PAIR ? 2 : variant == CMPS_VARIANT_WIDE ? 4 : 0);
    return CMPS_OK;
}

int cmps_psi_loss_bwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* grad_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || !h->fwd_saved || h->legacy)
        return fail(h, CMPS_ERR_STATE, "cmps_psi_loss_bwd: needs cmps_psi_loss_fwd(save_for_bwd=1) first");
    if (!audio_dev || !grad_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_loss_bwd: null pointer");
    if (B != h->saved_B || T != h->saved_T || audio_dev != h->saved_audio)
        return fail(h, CMPS_ERR_STATE, "cmps_psi_loss_bwd: audio / B / T differ from the forward call");
    Dev P = h->P;
    P.B = B;
    P.f16_shift = h->f16_shift;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // the stash layout belongs to the variant that wrote it
    KBind kb(h);
    if (h->saved_variant == CMPS_VARIANT_PAIR) {
        // reverse scan, then the gradient GEMM over the rows both scans left behind; one slab per PAIR of clips
        hipError_t e;
        { KScope ks("k_bwd_pair", s); e = launch_bwd_pair(P, audio_dev, s); }
        if (e == hipSuccess) { KScope ks("k_grad_gemm<1>", s); e = launch_grad_pair(P, audio_dev, s); }
        if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_loss_bwd (pair scan)");
        Dev Pp = P;
        Pp.B = (B + 1) / 2;
        KScope ks("reduce + finalize", s);
        e = launch_reduce_only(Pp, s);
        if (e == hipSuccess) e = launch_finalize_only(P, h->saved_loss, grad_dev, s);
        if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_loss_bwd (pair reduce)");
        return CMPS_OK;
    }
    if (h->saved_variant == CMPS_VARIANT_WIDE) {
        // float32 reverse scan, then the gradient GEMM (operands split into bf16 pieces on the fly); one slab per PAIR of clips
        hipError_t e;
        if (h->wide_chain == CMPS_WIDE_CHAIN_MFMA) { KScope ks("k_bwd_chain16", s); e = launch_bwd_chain16(P, audio_dev, s); }
        else { KScope ks("k_bwd_wide", s); e = launch_bwd_wide(P, audio_dev, s); }
        if (e == hipSuccess) {
            const int rm = h->rank1_mode == CMPS_RANK1_DEFAULT ? CMPS_RANK1_F16X2 : h->rank1_mode;
            KScope ks(rm == CMPS_RANK1_F16X2 ? "k_grad_gemm<f16x2>" : rm == CMPS_RANK1_BF16X2 ? "k_grad_gemm<2>" : "k_grad_gemm<3>", s);
            e = launch_grad_wide(P, audio_dev, rm == CMPS_RANK1_F16X2 ? -2 : rm == CMPS_RANK1_BF16X2 ? 2 : 3, s);
        }
        if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_loss_bwd (wide scan)");
        Dev Pp = P;
        Pp.B = (B + 1) / 2;
        KScope ks("reduce + finalize", s);
        e = launch_reduce_only(Pp, s);
        P.abar_fix = 1;              // the merged mat-vec: k_finalize removes the Q part of sum Re(u^dagger (Q + s R^dagger) ybar)
        if (e == hipSuccess) e = launch_finalize_only(P, h->saved_loss, grad_dev, s);
        if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_loss_bwd (wide reduce)");
        return CMPS_OK;
    }
    const bool wave = h->saved_variant == CMPS_VARIANT_WAVE || h->saved_variant == CMPS_VARIANT_WAVE32;
    const bool w16 = h->saved_variant == CMPS_VARIANT_WAVE && h->D <= 16;
    hipError_t e;
    {
        const bool two = wave && !w16 && h->bwd_waves == 2 && wave_rank1(h->rank1_mode) == CMPS_RANK1_F16X2 && h->f16_shift == 0;
        KScope ks(!wave ? "k_bwd_block" : w16 ? "k_bwd_wave16" : two ? "k_bwd_wave2w" : "k_bwd_wave", s);
        e = !wave ? launch_bwd_block(P, audio_dev, s) : w16 ? launch_bwd_wave16(P, audio_dev, s)
          : two ? launch_bwd_wave2w(P, audio_dev, s) : launch_bwd_wave(P, audio_dev, wave_rank1(h->rank1_mode), s);
    }
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_loss_bwd (scan)");
    P.abar_fix = wave ? 1 : 0;
    KScope ks("reduce + finalize", s);
    e = launch_reduce_finalize(P, h->saved_loss, grad_dev, s);
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_loss_bwd (reduce)");
    return CMPS_OK;
}

int cmps_psi_grad_status(cmps_handle_t h, int* sticky_out, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (sticky_out) *sticky_out = 0;
    if (!h->params_set || h->legacy || !h->P.status)
        return fail(h, CMPS_ERR_STATE, "cmps_psi_grad_status: needs cmps_set_params with a CMPS_WS_TRAIN workspace");
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned words[2] = {0u, 0u};
    hipError_t e = hipMemcpyAsync(words, h->P.status, sizeof words, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemsetAsync(h->P.status + 1, 0, sizeof(unsigned), s);        // the sticky word restarts
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_grad_status");
    if (sticky_out) *sticky_out = (int)words[1];
    if ((words[0] & 1u) && !(words[0] & 2u))
        return fail(h, CMPS_ERR_F16_RANGE,
                    "cmps_psi_grad_status: the gradient sums of the last cmps_psi_loss_bwd hold Inf / NaN although every per-clip loss is "
                    "finite: an fp16-split operand left its scaled range.  Fallback: CMPS_OPT_RANK1 = BF16X3, CMPS_OPT_WIDE_CHAIN = VALU, "
                    "then repeat cmps_psi_loss_fwd / _bwd");
    return CMPS_OK;
}

int cmps_psi_update_ancilla(cmps_handle_t h, const float* psi_in_dev, const float* signal_dev, float t,
                            int B, float* psi_out_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set) return fail(h, CMPS_ERR_STATE, "cmps_psi_update_ancilla: call cmps_set_params first");
    if (!psi_in_dev || !signal_dev || !psi_out_dev || B < 1)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_update_ancilla: bad argument");
    hipError_t e = launch_update_ancilla(h->P, psi_in_dev, signal_dev, t, B, psi_out_dev,
                                         static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_update_ancilla");
    return CMPS_OK;
}

int cmps_psi_states(cmps_handle_t h, int B, int T, float* psi_out_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || !h->fwd_saved)
        return fail(h, CMPS_ERR_STATE, "cmps_psi_states: needs cmps_psi_loss_fwd(save_for_bwd=1) first");
    if (!psi_out_dev || B != h->saved_B || T != h->saved_T)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_states: bad argument");
    hipError_t e = launch_states(h->P, B, psi_out_dev, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_states");
    return CMPS_OK;
}

int cmps_psi_sample(cmps_handle_t h, const float* noise_dev, int n, int length, float* out_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set) return fail(h, CMPS_ERR_STATE, "cmps_psi_sample: call cmps_set_params first");
    if (!noise_dev || !out_dev || n < 1 || length < 1)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_sample: bad argument");
    if (length > h->L.N)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_psi_sample: length exceeds T - 1 of cmps_set_params (the per-step tables)");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int sv = resolve_variant(h);
    // wave-per-path kernel for D <= 32, the wide chain's sampling mode for 32 < D <= 128 (float32; also what the bf16 pair variant
    // samples with: sampling has no reduced-precision form), the block kernel otherwise
    hipError_t e = (sv == CMPS_VARIANT_WAVE || sv == CMPS_VARIANT_WAVE32) ? launch_sample_wave(h->P, noise_dev, n, length, out_dev, s)
                   : (sv == CMPS_VARIANT_WIDE || sv == CMPS_VARIANT_PAIR)  ? launch_sample_wide(h->P, noise_dev, n, length, out_dev, s)
                                                                           : launch_sample_block(h->P, noise_dev, n, length, out_dev, s);
    if (e != hipSuccess) return fail_hip(h, e, "cmps_psi_sample");
    return CMPS_OK;
}

// ---------------------------------------------------------------------------------------------------
// legacy AudioMPS arithmetic
// ---------------------------------------------------------------------------------------------------
int cmps_legacy_set_params(cmps_handle_t h, const float* R_dev, const float* Q_re_dev, const float* Q_im_dev,
                           double delta_t, int T, int B_max, int flags, void* workspace_dev, size_t workspace_bytes,
                           void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!R_dev || !Q_re_dev || !Q_im_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_legacy_set_params: null parameter pointer");
    if (T < 2 || B_max < 1) return fail(h, CMPS_ERR_BAD_ARG, "cmps_legacy_set_params: need T >= 2 and B_max >= 1");
    if (!workspace_dev) return fail(h, CMPS_ERR_WORKSPACE, "cmps_legacy_set_params: null workspace");
    Layout L = make_layout(h->D, B_max, T, flags);
    if (workspace_bytes < L.total) return fail(h, CMPS_ERR_WORKSPACE, "cmps_legacy_set_params: workspace too small");
    if (((uintptr_t)workspace_dev & 255) != 0)
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_legacy_set_params: workspace must be 256-byte aligned");
    char* ws = static_cast<char*>(workspace_dev);
    Dev P{};
    P.D = L.D; P.DP = L.DP; P.B = B_max; P.T = T; P.N = L.N;
    P.R = reinterpret_cast<float2*>(ws + L.off_R);
    P.RT = reinterpret_cast<float2*>(ws + L.off_RT);
    P.Q = reinterpret_cast<float2*>(ws + L.off_Q);
    P.QT = reinterpret_cast<float2*>(ws + L.off_QT);
    P.stash = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float2*>(ws + L.off_stash) : nullptr;
    P.hst = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_hst) : nullptr;
    P.scal = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_scal) : nullptr;
    P.slabs = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_slabs) : nullptr;
    P.sums = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<float*>(ws + L.off_sums) : nullptr;
    P.status = (flags & CMPS_WS_TRAIN) ? reinterpret_cast<unsigned*>(ws + L.off_status) : nullptr;
    P.gops = ((flags & CMPS_WS_TRAIN) && L.D > 32) ? static_cast<void*>(ws + L.off_gops) : nullptr;      // the wide kernels' ybar rows
    P.opmax = ((flags & CMPS_WS_TRAIN) && L.D > 32) ? reinterpret_cast<float*>(ws + L.off_opmax) : nullptr;
    P.freqs = reinterpret_cast<float*>(ws + L.off_freqs);
    P.qflag = reinterpret_cast<unsigned*>(ws + L.off_qflag);
    P.slab_floats = L.slab_floats;
    P.dt = (float)delta_t;
    // the tables of the pure-state wave kernels, which the D <= 32 legacy kernels share (cmps_wave2.hip / cmps_wave.hip, LEGACY):
    // rotation rho = 1, psi_0 = e_0, no time table
    P.psi0 = reinterpret_cast<float2*>(ws + L.off_psi0);
    P.dtk = reinterpret_cast<float*>(ws + L.off_dtk);
    P.rho = reinterpret_cast<float2*>(ws + L.off_rho);
    P.A = 1.0f;
    hipError_t e = launch_pack_legacy(P, R_dev, Q_re_dev, Q_im_dev, const_cast<float2*>(P.R), const_cast<float2*>(P.RT),
                                      const_cast<float2*>(P.Q), const_cast<float2*>(P.QT), static_cast<hipStream_t>(stream));
    if (e == hipSuccess) e = launch_legacy_tables(P, const_cast<float2*>(P.psi0), const_cast<float*>(P.dtk), const_cast<float2*>(P.rho),
                                                  static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_legacy_set_params");
    h->L = L; h->P = P; h->ws = ws;
    h->tt_ws = nullptr;               // the time table of the PsiCMPS mode is no longer valid for this workspace
    h->params_set = true;
    h->legacy = true;
    h->fwd_saved = false;
    h->rho_set = false;
    return CMPS_OK;
}

int cmps_legacy_loss_fwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* loss_dev, int save_for_bwd,
                         void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || !h->legacy) return fail(h, CMPS_ERR_STATE, "cmps_legacy_loss_fwd: call cmps_legacy_set_params first");
    if (!audio_dev || !loss_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_legacy_loss_fwd: null pointer");
    if (T != h->L.T || B < 1 || B > h->L.B) return fail(h, CMPS_ERR_BAD_ARG, "cmps_legacy_loss_fwd: shape differs from set_params");
    if (save_for_bwd && !(h->L.flags & CMPS_WS_TRAIN))
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_legacy_loss_fwd: save_for_bwd needs a CMPS_WS_TRAIN workspace");
    Dev P = h->P;
    P.B = B;
    // D <= 32: the pure-state wave kernels in LEGACY mode (cmps_wave2.hip, cmps_wave.hip); above: the wide kernels in LEGACY mode
    // (cmps_wide.hip; round 5); CMPS_VARIANT_BLOCK: the general one-workgroup-per-clip kernels (cmps_legacy.hip), the cross-check
    const bool wave = h->D <= 32 && h->variant_req != CMPS_VARIANT_BLOCK;
    const bool wide = h->D > 32 && h->variant_req != CMPS_VARIANT_BLOCK;
    const bool f16 = h->rank1_mode == CMPS_RANK1_DEFAULT || h->rank1_mode == CMPS_RANK1_F16X2;
    KBind kb(h);
    hipError_t e = wave ? launch_fwd_legacy_wave(P, audio_dev, loss_dev, save_for_bwd != 0, static_cast<hipStream_t>(stream))
                 : wide ? launch_fwd_wide_legacy(P, audio_dev, loss_dev, save_for_bwd != 0, f16, static_cast<hipStream_t>(stream))
                        : launch_fwd_legacy(P, audio_dev, loss_dev, save_for_bwd != 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_legacy_loss_fwd");
    h->saved_variant = wave ? CMPS_VARIANT_WAVE : wide ? CMPS_VARIANT_WIDE : CMPS_VARIANT_BLOCK;
    h->fwd_saved = save_for_bwd != 0;
    h->saved_B = B; h->saved_T = T; h->saved_audio = audio_dev; h->saved_loss = loss_dev;
    return CMPS_OK;
}

int cmps_legacy_loss_bwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* grad_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || !h->legacy || !h->fwd_saved)
        return fail(h, CMPS_ERR_STATE, "cmps_legacy_loss_bwd: needs cmps_legacy_loss_fwd(save_for_bwd=1) first");
    if (!audio_dev || !grad_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_legacy_loss_bwd: null pointer");
    if (B != h->saved_B || T != h->saved_T || audio_dev != h->saved_audio)
        return fail(h, CMPS_ERR_STATE, "cmps_legacy_loss_bwd: audio / B / T differ from the forward call");
    Dev P = h->P;
    P.B = B;
    hipStream_t s = static_cast<hipStream_t>(stream);
    KBind kb(h);
    hipError_t e;
    Dev Pr = P;                                                    // what the slab reduction runs over
    if (h->saved_variant == CMPS_VARIANT_WIDE) {
        // reverse chain, then the gradient GEMM over the rows both scans left behind; one slab per PAIR of clips
        { KScope ks("k_bwd_wide<legacy>", s); e = launch_bwd_wide_legacy(P, audio_dev, s); }
        if (e == hipSuccess) {
            KScope ks("k_grad_gemm<legacy>", s);
            e = launch_grad_wide_legacy(P, audio_dev, h->rank1_mode == CMPS_RANK1_DEFAULT || h->rank1_mode == CMPS_RANK1_F16X2, s);
        }
        Pr.B = (B + 1) / 2;
    } else {
        e = h->saved_variant == CMPS_VARIANT_WAVE ? launch_bwd_legacy_wave(P, audio_dev, wave_rank1(h->rank1_mode), s) : launch_bwd_legacy(P, audio_dev, s);
    }
    if (e != hipSuccess) return fail_hip(h, e, "cmps_legacy_loss_bwd (scan)");
    e = launch_reduce_only(Pr, s);
    if (e == hipSuccess) e = launch_finalize_legacy(P, h->saved_loss, grad_dev, s);
    if (e != hipSuccess) return fail_hip(h, e, "cmps_legacy_loss_bwd (reduce)");
    return CMPS_OK;
}

// ---------------------------------------------------------------------------------------------------
// RhoCMPS (model.py:55-203)
// ---------------------------------------------------------------------------------------------------
size_t cmps_rho_workspace_bytes(int D, int rank, int B, int T, int flags) {
    if (D < 1 || D > 128 || rank < 1 || B < 1 || T < 2) return 0;
    return make_rho_layout(D, rank, B, T, flags).total;
}

int cmps_rho_set_state(cmps_handle_t h, const float* phi_re_dev, const float* phi_im_dev, int rank, int T, int B_max,
                       int flags, void* rho_workspace_dev, size_t rho_workspace_bytes, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || h->legacy) return fail(h, CMPS_ERR_STATE, "cmps_rho_set_state: call cmps_set_params first");
    if (!phi_re_dev || !phi_im_dev || rank < 1 || B_max < 1 || T < 2)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_set_state: bad argument");
    if (T > h->L.T) return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_set_state: T exceeds T of cmps_set_params (the per-step tables)");
    if (rank > 128) return fail(h, CMPS_ERR_UNSUPPORTED_D, "cmps_rho_set_state: rank above 128");
    const size_t rD = (size_t)rank * h->D;
    if (!rho_workspace_dev) return fail(h, CMPS_ERR_WORKSPACE, "cmps_rho_set_state: null workspace");
    if (((uintptr_t)rho_workspace_dev & 255) != 0)
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_rho_set_state: workspace must be 256-byte aligned");
    RhoLayout RL = make_rho_layout(h->D, rank, B_max, T, flags);
    if (rho_workspace_bytes < RL.total) {
        char buf[160];
        snprintf(buf, sizeof buf, "cmps_rho_set_state: workspace has %zu bytes, needs %zu", rho_workspace_bytes, RL.total);
        return fail(h, CMPS_ERR_WORKSPACE, buf);
    }
    char* ws = static_cast<char*>(rho_workspace_dev);
    RhoDev W{};
    W.rank = rank;
    W.phi0 = reinterpret_cast<float2*>(ws + RL.off_phi0);
    const bool train = (flags & CMPS_WS_TRAIN) != 0;
    W.stash = train ? reinterpret_cast<float2*>(ws + RL.off_stash) : nullptr;
    W.scal = train ? reinterpret_cast<float*>(ws + RL.off_scal) : nullptr;
    W.p1 = (train && h->D <= 32) ? reinterpret_cast<float*>(ws + RL.off_p1) : nullptr;
    W.stash_layout = 0;
    W.cols = rD > RHO_LDS_COLS_MAX ? reinterpret_cast<float2*>(ws + RL.off_cols) : nullptr;
    W.cols_blocks = B_max;
    W.slabs = train ? reinterpret_cast<float*>(ws + RL.off_slabs) : nullptr;
    W.sums = train ? reinterpret_cast<float*>(ws + RL.off_sums) : nullptr;
    W.slab_floats = RL.slab_floats;
    W.wslabs = (train && h->D <= 32) ? reinterpret_cast<float*>(ws + RL.off_wslabs) : nullptr;
    W.wsums = (train && h->D <= 32) ? reinterpret_cast<float*>(ws + RL.off_wsums) : nullptr;
    W.vrank = RL.vrank;                                          // > 0: the sections of the wide (virtual-clip) path exist
    if (RL.vrank) {
        W.vphi = reinterpret_cast<float2*>(ws + RL.off_vphi);
        W.vstash = reinterpret_cast<float*>(ws + RL.off_vstash);
        W.vgops = reinterpret_cast<float*>(ws + RL.off_vgops);
        W.vopmax = reinterpret_cast<float*>(ws + RL.off_vopmax);
        W.vscal = reinterpret_cast<float*>(ws + RL.off_vscal);
        W.rscal = reinterpret_cast<float*>(ws + RL.off_rscal);
        W.vslabs = reinterpret_cast<float*>(ws + RL.off_vslabs);
        W.vsums = reinterpret_cast<float*>(ws + RL.off_vsums);
        W.vaudio = reinterpret_cast<float*>(ws + RL.off_vaudio);
        W.gphi = reinterpret_cast<float*>(ws + RL.off_gphi);
        W.vslab_floats = RL.vslab_floats;
    }
    hipError_t e = launch_pack_phi(h->P, W, phi_re_dev, phi_im_dev, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_rho_set_state");
    h->RL = RL; h->W = W;
    h->rho_B = B_max; h->rho_T = T; h->rho_flags = flags;
    h->rho_set = true;
    h->rho_saved = false;
    h->rho_bwd_ok = false;
    return CMPS_OK;
}

int cmps_rho_loss_fwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* loss_dev, int save_for_bwd,
                      void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || h->legacy || !h->rho_set)
        return fail(h, CMPS_ERR_STATE, "cmps_rho_loss_fwd: call cmps_set_params and cmps_rho_set_state first");
    if (!audio_dev || !loss_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_loss_fwd: null pointer");
    if (T != h->rho_T) return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_loss_fwd: T differs from cmps_rho_set_state");
    if (B < 1 || B > h->rho_B) return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_loss_fwd: B outside [1, B_max]");
    if (save_for_bwd && !(h->rho_flags & CMPS_WS_TRAIN))
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_rho_loss_fwd: save_for_bwd needs a CMPS_WS_TRAIN rho workspace");
    Dev P = h->P;
    P.B = B; P.T = T; P.N = T - 1;
    // D <= 32 (and rank <= 32): one wavefront per clip, unless the block variant was asked for -- the forward as row-array
    // GEMMs on the matrix cores (cmps_rho_mfma.hip); CMPS_VARIANT_WAVE32 keeps the column-by-column kernel (cross-check)
    const bool wave = h->D <= 32 && h->W.rank <= 32 && h->variant_req != CMPS_VARIANT_BLOCK;
    // the row-array GEMM forward costs the same at every rank; with the virtual-clip reverse sweep behind it (3.75 + 0.11 rank ms at T = 1000,
    // B = 256 against 1.65 rank ms for the column kernels: scripts/bench_next_rows.py) it wins from rank 3, with k_bwd_rho_mfma from rank 9
    const bool mfma = wave && h->variant_req != CMPS_VARIANT_WAVE32 && h->W.rank > (h->rho_virtual_bwd && save_for_bwd ? 2 : 8);
    // 32 < D <= 128, training forward (round 5): the columns as virtual clips of the wide kernels (cmps_wide.hip), when the rho workspace
    // has the sections for it (cmps_rho_workspace_bytes: CMPS_WS_TRAIN and the column vectors fit the LDS); else the general kernels
    const bool wide = h->D > 32 && h->W.vrank > 0 && save_for_bwd != 0 && h->variant_req != CMPS_VARIANT_BLOCK;
    const bool f16 = h->rank1_mode == CMPS_RANK1_DEFAULT || h->rank1_mode == CMPS_RANK1_F16X2;
    hipStream_t s = static_cast<hipStream_t>(stream);
    KBind kb(h);
    hipError_t e = wide ? launch_fwd_rho_wide(P, h->W, audio_dev, loss_dev, f16, s)
                 : mfma ? launch_fwd_rho_mfma(P, h->W, audio_dev, loss_dev, save_for_bwd != 0, f16, !h->rho_virtual_bwd, s)
                 : wave ? launch_fwd_rho_wave(P, h->W, audio_dev, loss_dev, save_for_bwd != 0, s)
                        : launch_fwd_rho(P, h->W, audio_dev, loss_dev, save_for_bwd != 0, s);
    if (e != hipSuccess) return fail_hip(h, e, "cmps_rho_loss_fwd");
    h->W.stash_layout = wide ? 3 : mfma ? 2 : (wave ? 1 : 0);
    h->rho_fwd_grad1 = mfma && !h->rho_virtual_bwd;               // the forward's part of Rbar exists (k_bwd_rho_mfma needs it)
    h->rho_saved = h->rho_bwd_ok = save_for_bwd != 0;
    h->rho_saved_B = B; h->rho_saved_steps = T - 1;
    h->saved_audio = audio_dev; h->saved_loss = loss_dev;
    return CMPS_OK;
}

int cmps_rho_loss_bwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* grad_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || !h->rho_set || !h->rho_saved || !h->rho_bwd_ok)
        return fail(h, CMPS_ERR_STATE, "cmps_rho_loss_bwd: needs cmps_rho_loss_fwd(save_for_bwd=1) first");
    if (!audio_dev || !grad_dev) return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_loss_bwd: null pointer");
    if (B != h->rho_saved_B || T - 1 != h->rho_saved_steps || audio_dev != h->saved_audio)
        return fail(h, CMPS_ERR_STATE, "cmps_rho_loss_bwd: audio / B / T differ from the forward call");
    Dev P = h->P;
    P.B = B; P.T = T; P.N = T - 1;
    P.slabs = h->W.slabs; P.sums = h->W.sums; P.slab_floats = h->W.slab_floats;   // the reduction runs on the rho slabs
    hipStream_t s = static_cast<hipStream_t>(stream);
    KBind kb(h);
    if (h->W.stash_layout == 3) {                                 // the wide kernels on virtual clips: reverse chain, GEMM, reduction, closing terms
        const int rm = h->rank1_mode == CMPS_RANK1_DEFAULT ? CMPS_RANK1_F16X2 : h->rank1_mode;
        const hipError_t ew = launch_bwd_rho_wide(P, h->W, h->saved_loss, grad_dev,
                                                  rm == CMPS_RANK1_F16X2 ? -2 : rm == CMPS_RANK1_BF16X2 ? 2 : 3, s);
        if (ew != hipSuccess) return fail_hip(h, ew, "cmps_rho_loss_bwd (wide)");
        return CMPS_OK;
    }
    if (h->W.stash_layout == 2 && !h->rho_virtual_bwd && !h->rho_fwd_grad1)
        return fail(h, CMPS_ERR_STATE, "cmps_rho_loss_bwd: CMPS_OPT_RHO_BWD changed between the forward and the reverse call");
    if (h->W.stash_layout == 2 && h->rho_virtual_bwd) {
        // the row-array forward's rows through the pure-state wave reverse scan, one virtual clip per column (cmps_rho_wave.hip)
        const hipError_t ew = launch_bwd_rho_virtual_wave(P, h->W, audio_dev, h->saved_loss, grad_dev, wave_rank1(h->rank1_mode), h->bwd_waves, s);
        if (ew != hipSuccess) return fail_hip(h, ew, "cmps_rho_loss_bwd (virtual clips)");
        return CMPS_OK;
    }
    hipError_t e = h->W.stash_layout == 2 ? launch_bwd_rho_mfma(P, h->W, audio_dev, s)
                 : h->W.stash_layout == 1 ? launch_bwd_rho_wave(P, h->W, audio_dev, s) : launch_bwd_rho(P, h->W, audio_dev, s);
    if (e != hipSuccess) return fail_hip(h, e, "cmps_rho_loss_bwd (scan)");
    P.abar_fix = h->W.stash_layout == 2 ? 1 : 0;   // the MFMA scan sums Re(u^dagger (Q + s R^dagger) ybar); k_finalize removes the Q part
    e = launch_reduce_finalize(P, h->saved_loss, grad_dev, s);
    if (e == hipSuccess) e = launch_finalize_rho(P, h->W, grad_dev, s);
    if (e != hipSuccess) return fail_hip(h, e, "cmps_rho_loss_bwd (reduce)");
    return CMPS_OK;
}

int cmps_rho_update_ancilla(cmps_handle_t h, const float* rho_in_dev, const float* signal_dev, float t, int B,
                            float* rho_out_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || h->legacy) return fail(h, CMPS_ERR_STATE, "cmps_rho_update_ancilla: call cmps_set_params first");
    if (!rho_in_dev || !signal_dev || !rho_out_dev || B < 1 || rho_in_dev == rho_out_dev)
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_update_ancilla: bad argument");
    hipError_t e = launch_update_ancilla_rho(h->P, rho_in_dev, signal_dev, t, B, rho_out_dev, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_rho_update_ancilla");
    return CMPS_OK;
}

int cmps_rho_sample(cmps_handle_t h, const float* noise_dev, int n, int length, float* out_dev, int save_states,
                    void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || h->legacy || !h->rho_set)
        return fail(h, CMPS_ERR_STATE, "cmps_rho_sample: call cmps_set_params and cmps_rho_set_state first");
    if (!noise_dev || !out_dev || n < 1 || length < 1) return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_sample: bad argument");
    if (length > h->L.N) return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_sample: length exceeds T - 1 of cmps_set_params");
    // (only when the SAMPLER's three column arrays exceed the LDS: the workspace section exists from a smaller rank * D on, for the
    // reverse scan's four -- ADVICE r3)
    const bool mfma_sampler = h->D <= 32 && h->W.rank <= 32 && h->variant_req != CMPS_VARIANT_BLOCK;
    if (!mfma_sampler && rho_cols_spill(3, h->W.rank, h->P.D) && n > h->rho_B)
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_rho_sample: at this rank * D the columns live in the workspace: n must not exceed B_max");
    if (save_states && (!(h->rho_flags & CMPS_WS_TRAIN) || (size_t)n * length > (size_t)h->rho_B * (h->rho_T - 1)))
        return fail(h, CMPS_ERR_WORKSPACE, "cmps_rho_sample: save_states needs a CMPS_WS_TRAIN rho workspace with B_max*(T-1) >= n*length");
    // D <= 32 (rank <= 32): the row-array GEMM sampler, one wavefront per path (cmps_rho_mfma.hip); CMPS_VARIANT_BLOCK keeps the
    // general workgroup-per-path kernel (cross-check; 93 us per step at rank 32 against 2 us)
    const bool mfma = mfma_sampler;
    hipError_t e = mfma ? launch_sample_rho_mfma(h->P, h->W, noise_dev, n, length, out_dev, save_states != 0,
                                                 h->rank1_mode == CMPS_RANK1_DEFAULT || h->rank1_mode == CMPS_RANK1_F16X2, static_cast<hipStream_t>(stream))
                        : launch_sample_rho(h->P, h->W, noise_dev, n, length, out_dev, save_states != 0, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_rho_sample");
    h->rho_saved = save_states != 0;
    h->W.stash_layout = mfma ? 2 : 0;
    h->rho_bwd_ok = false;
    h->rho_saved_B = n; h->rho_saved_steps = length;
    return CMPS_OK;
}

int cmps_rho_states(cmps_handle_t h, int B, int steps, float* rho_out_dev, float* purity_out_dev, void* stream) {
    if (!h) return CMPS_ERR_BAD_ARG;
    if (!h->params_set || !h->rho_set || !h->rho_saved)
        return fail(h, CMPS_ERR_STATE, "cmps_rho_states: needs cmps_rho_loss_fwd(save_for_bwd=1) or cmps_rho_sample(save_states=1) first");
    if (B != h->rho_saved_B || steps != h->rho_saved_steps || (!rho_out_dev && !purity_out_dev))
        return fail(h, CMPS_ERR_BAD_ARG, "cmps_rho_states: bad argument");
    hipError_t e = launch_states_rho(h->P, h->W, B, steps, rho_out_dev, purity_out_dev, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(h, e, "cmps_rho_states");
    return CMPS_OK;
}

// ---------------------------------------------------------------------------------------------------
// per-kernel durations recorded while CMPS_OPT_KERNEL_EVENTS is on
// ---------------------------------------------------------------------------------------------------
int cmps_kernel_times(cmps_handle_t h, char* names, size_t names_bytes, float* ms_sum, int* calls, int cap) {
    if (!h) return -1;
    if (!h->ktimer) { fail(h, CMPS_ERR_STATE, "cmps_kernel_times: CMPS_OPT_KERNEL_EVENTS is off"); return -1; }
    if (!names || !ms_sum || !calls || cap < 1 || names_bytes < 1) { fail(h, CMPS_ERR_BAD_ARG, "cmps_kernel_times: bad argument"); return -1; }
    KTimer* t = h->ktimer;
    std::vector<const char*> order;
    std::vector<double> sum;
    std::vector<int> cnt;
    for (auto& r : t->recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) { fail(h, CMPS_ERR_HIP, "cmps_kernel_times: hipEventSynchronize"); return -1; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.a, r.b);
        size_t i = 0;
        for (; i < order.size(); ++i) if (!strcmp(order[i], r.name)) break;
        if (i == order.size()) { order.push_back(r.name); sum.push_back(0.0); cnt.push_back(0); }
        sum[i] += ms; cnt[i] += 1;
        t->pool.push_back(r.a); t->pool.push_back(r.b);
    }
    t->recs.clear();
    std::string joined;
    int n = 0;
    for (size_t i = 0; i < order.size() && n < cap; ++i, ++n) {
        if (i) joined += '\n';
        joined += order[i];
        ms_sum[n] = (float)sum[i];
        calls[n] = cnt[i];
    }
    if (joined.size() + 1 > names_bytes) { fail(h, CMPS_ERR_BAD_ARG, "cmps_kernel_times: names buffer too small"); return -1; }
    memcpy(names, joined.c_str(), joined.size() + 1);
    return n;
}

// ---------------------------------------------------------------------------------------------------
// host utility of the data-format row: CRC-32C (Castagnoli) of TFRecord framing
// ---------------------------------------------------------------------------------------------------
unsigned cmps_crc32c(const void* data, size_t n, unsigned crc_in) {
    const unsigned char* p = static_cast<const unsigned char*>(data);
    unsigned long long crc = crc_in ^ 0xFFFFFFFFu;
#if defined(__SSE4_2__)
    while (n && ((uintptr_t)p & 7)) { crc = __builtin_ia32_crc32qi((unsigned)crc, *p++); --n; }
    for (; n >= 8; n -= 8, p += 8) {
        unsigned long long v;
        memcpy(&v, p, 8);
        crc = __builtin_ia32_crc32di(crc, v);
    }
    while (n--) crc = __builtin_ia32_crc32qi((unsigned)crc, *p++);
#else
    static unsigned table[256];
    static bool init = false;
    if (!init) {
        for (unsigned i = 0; i < 256; ++i) {
            unsigned c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    while (n--) crc = table[(crc ^ *p++) & 0xFF] ^ (crc >> 8);
#endif
    return (unsigned)crc ^ 0xFFFFFFFFu;
}

}  // extern "C"
