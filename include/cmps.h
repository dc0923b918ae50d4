/*
 * cmps.h -- C ABI of libcmps.so: the MI355X (gfx950) implementation of audio-mps's per-timestep cMPS
 * contraction scan (PsiCMPS log-likelihood forward and its gradient).
 *
 * The reference has no FFI: the path sits behind Python object attributes (PsiCMPS(...).loss,
 * /root/reference/model.py:211-225, 257-334) and its gradient is produced by TensorFlow's autodiff
 * (train.py:89).  Each entry point below names the reference lines it replaces.  All pointers named
 * *_dev are DEVICE pointers owned by the caller; the library allocates no device memory, launches
 * everything asynchronously on the caller's stream and never synchronises.  `stream` is a hipStream_t
 * passed as void* (NULL = the default stream).  A handle must not be shared between host threads.
 *
 * Every function returns CMPS_OK (0) or a CMPS_ERR_* code; cmps_last_error(h) gives the message.
 * NaN / Inf in the loss is NOT an error at this boundary (it propagates, as in the reference where
 * only tests/test_model.py:113 looks at it).
 */
#ifndef CMPS_H_
#define CMPS_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct cmps_handle_s* cmps_handle_t;

enum {
    CMPS_OK = 0,
    CMPS_ERR_BAD_ARG = 1,       /* null pointer, non-positive size, T < 2 ... */
    CMPS_ERR_UNSUPPORTED_D = 2, /* bond dimension outside [1, 128] */
    CMPS_ERR_WORKSPACE = 3,     /* workspace missing or smaller than cmps_workspace_bytes() */
    CMPS_ERR_HIP = 4,           /* a HIP runtime call or kernel launch failed */
    CMPS_ERR_STATE = 5,         /* call order: set_params -> fwd(save_for_bwd=1) -> bwd */
    CMPS_ERR_F16_RANGE = 6      /* cmps_psi_grad_status only: the gradient sums of the last cmps_psi_loss_bwd hold Inf / NaN although
                                 * every per-clip loss is finite -- an fp16-split operand left its scaled range (see CMPS_RANK1_F16X2) */
};

/* workspace flags */
enum {
    CMPS_WS_FWD_ONLY = 0, /* loss only */
    CMPS_WS_TRAIN = 1,    /* forward + backward (adds the per-step state stash and gradient slabs) */
    CMPS_WS_REUSE_TABLES = 4, /* OR-ed into `flags` of cmps_set_params: the caller vouches that the workspace still holds what the
                           * handle's previous cmps_set_params left in it -- the time table is then rebuilt only when T or delta_t
                           * changed.  WITHOUT this flag every call rebuilds every table (the safe default for a C caller). */
    CMPS_WS_FRESH = 2     /* (kept for v200 callers; now the default behaviour) the workspace memory was (re)allocated, zeroed or used
                           * for something else since the previous call -- rebuild every cached table (see below) */
};

/* Kernel variants (cmps_set_variant).  AUTO picks the register-resident wave-per-clip kernels for D <= 32 and the "wide"
 * kernels above that.  Both follow the reference's float32 / complex64 arithmetic (model.py:300-325) in this sense:
 *   - state, identity part of every update, normalisation, loss and every accumulation are float32;
 *   - D <= 32: the mat-vec on the serial chain is an fp32 FMA chain.  The two products nothing waits for run on the matrix
 *     cores with split operands and fp32 accumulation: the loss product H y (three bf16 pieces, six products) and the rank-1
 *     gradient sums (CMPS_OPT_RANK1; default two scaled fp16 pieces, three products);
 *   - 32 < D <= 128: the mat-vecs of both serial chains, H y and the gradient GEMM all run on the matrix cores with two
 *     scaled fp16 pieces per operand (CMPS_OPT_WIDE_CHAIN, CMPS_OPT_RANK1 select the fp32-VALU / bf16-piece forms).
 * Split products carry 22-24 operand bits: within float32 rounding of an fp32 FMA chain, not bit-identical to it.
 * Only CMPS_VARIANT_BLOCK is plain fp32 FMA code throughout. */
enum {
    CMPS_VARIANT_AUTO = 0,
    CMPS_VARIANT_BLOCK = 1, /* one workgroup per clip, any D <= 128 */
    CMPS_VARIANT_WAVE = 2,  /* wavefront-per-clip kernels, state and R in registers, D <= 32: the 16-row lane layout
                             * (cmps_wave16.hip) for D <= 16, the 32-row layout above that */
    CMPS_VARIANT_PAIR = 3,  /* 32 < D <= 128: one workgroup per pair of clips, matrices as bf16 MFMA fragments, fp32 accumulate */
    CMPS_VARIANT_WAVE32 = 4, /* the 32-row wave layout for every D <= 32 (zero padding below 32; cross-check of the 16-row layout) */
    CMPS_VARIANT_WIDE = 5   /* 32 < D <= 128 in float32 (what AUTO picks there): one workgroup per pair of clips, R / Q resident in
                             * registers for the whole launch; arithmetic selected by CMPS_OPT_WIDE_CHAIN and CMPS_OPT_RANK1 */
};

/* options (cmps_set_option / cmps_get_option) */
enum {
    CMPS_OPT_RANK1 = 1 /* arithmetic of the rank-1 gradient sums: the wave-per-clip reverse scan of 17 <= D <= 32 (the 16-row layout
                        * of D <= 16 always uses exact fp32 MFMAs) and the gradient GEMM of the wide kernels (32 < D <= 128:
                        * BF16X2 = two bf16 pieces / three products, F16X2 = two fp16 pieces / three products, anything else = three
                        * bf16 pieces / six products) */,
    CMPS_OPT_WIDE_CHAIN = 3 /* how the wide kernels (32 < D <= 128, float32) run the training forward's serial chain: see the values below */,
    CMPS_OPT_RHO_BWD = 5 /* which reverse sweep follows the RhoCMPS row-array GEMM forward (D <= 32; the forward runs from rank 3 with
                        * CMPS_RHO_BWD_VIRTUAL, from rank 9 with CMPS_RHO_BWD_GEMM, the column-by-column kernels below that): see the values below */,
    CMPS_OPT_BWD_WAVES = 6 /* the reverse scan of 17 <= D <= 32 (PsiCMPS arithmetic, F16X2 sums): 2 (default) = two wavefronts per clip on one
                        * SIMD, a chain wave and a gradient wave (k_bwd_wave2w, cmps_wave_bwd2.hip); 1 = one wavefront per clip (k_bwd_wave,
                        * the round-4 kernel: the A/B setting).  Same float32 tolerances; every other CMPS_OPT_RANK1 value, a non-zero
                        * CMPS_OPT_F16_SCALE_SHIFT and the legacy mode run one wavefront whatever this says; the RhoCMPS reverse sweep on virtual clips
                        * (CMPS_OPT_RHO_BWD) follows it */,
    CMPS_OPT_F16_SCALE_SHIFT = 4 /* DIAGNOSTIC, default 0: added to the exponent of every data-dependent fp16 scale of the wave reverse
                        * scan's F16X2 arithmetic (range -40 .. 40).  A positive value pushes the pieces out of fp16 range on purpose:
                        * how tests/test_gpu_parity.py provokes CMPS_ERR_F16_RANGE.  No reference counterpart. */,
    CMPS_OPT_KERNEL_EVENTS = 2 /* 1: every kernel cmps_psi_loss_fwd / _bwd launch is bracketed by two HIP events on the caller's stream
                        * (read and reset with cmps_kernel_times); 0 (default): nothing is recorded.  A measurement aid -- the reference
                        * has no counterpart (no tracing / profiling hooks); bench.py uses it OUTSIDE its timed region to price
                        * each kernel of a multi-kernel family against the pipe it runs on */
};
/* Values of CMPS_OPT_RANK1.  All accumulate in fp32; they differ in how the two factors of every product
 * dR += a b^dagger enter the matrix cores:
 *   EXACT_F32  v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fma chain (slowest: it holds the fp32 ALUs).
 *   BF16X2     each factor split into two bf16 pieces, 3 products: 16 operand bits, error <= ~2^-16 |a||b|.
 *   BF16X3     each factor split EXACTLY into three bf16 pieces (8+8+8 bits), 6 products: 24 operand bits,
 *              error <= 2^-23 |a||b| (what is dropped is below the fp32 rounding of the product).
 *   F16X2      each factor multiplied by a power of two and split into two fp16 pieces, round to nearest
 *              (11+1+11+1 bits), 3 products on v_mfma_f32_32x32x16_f16 / 16x16x32_f16: BF16X2's instruction count in
 *              BF16X3's accuracy class.  Error <= ~2^-22 |a||b| for factors within 2^-18 of the largest value their
 *              scale was chosen for, and <= 2^-39 of that largest value below.
 * Which kernels know which value, and where the fp16 scales come from:
 *   kernel                                        EXACT_F32  BF16X2  BF16X3  F16X2                          DEFAULT means
 *   wave reverse scan, 17 <= D <= 32 (PsiCMPS)    yes        yes     yes     scales per eight-step octet    F16X2
 *   wave reverse scan in legacy AudioMPS mode     yes        yes     yes     runs BF16X3                    BF16X3
 *   wave kernels, D <= 16                         always exact fp32 MFMAs (the option is ignored)
 *   wide kernels' GEMMs (H y, gradient), D > 32   runs BF16X3  yes   yes     scales per pair of clips       F16X2
 *   RhoCMPS row-array GEMM forward and sampler    runs BF16X3 (every value but F16X2 / DEFAULT)  scales per clip / fixed   F16X2
 *   RhoCMPS reverse scan, pair (bf16) kernels     the option is ignored
 * The fp16 scales follow guaranteed bounds; should one ever be violated the pieces overflow to Inf and the gradient
 * comes out non-finite: cmps_psi_grad_status reports that as CMPS_ERR_F16_RANGE. */
enum {
    CMPS_RANK1_EXACT_F32 = 0,
    CMPS_RANK1_BF16X2 = 1,
    CMPS_RANK1_BF16X3 = 2,
    CMPS_RANK1_F16X2 = 3,
    CMPS_RANK1_DEFAULT = 4   /* a new handle's setting: see the table above (the cheapest arithmetic of the 24-operand-bit
                              * class each kernel has; scripts/rank1_accuracy_wide.py,
                              * tests/test_gpu_parity.py::test_rank1_modes_order_of_accuracy) */
};

/* values of CMPS_OPT_RHO_BWD */
enum {
    CMPS_RHO_BWD_VIRTUAL = 0,  /* (a new handle's setting) every column of rho as one clip of the pure-state wave reverse scan (k_bwd_wave): given
                                * the clip's per-step scalars the column cotangents do not couple; cost linear in the rank, rank-1 sums in the
                                * CMPS_OPT_RANK1 arithmetic */
    CMPS_RHO_BWD_GEMM = 1      /* k_bwd_rho_mfma: the cotangent array as row-array GEMMs (bf16 x 3), the same cost at every rank <= 32 */
};

/* values of CMPS_OPT_WIDE_CHAIN */
enum {
    CMPS_WIDE_CHAIN_VALU = 0,   /* k_fwd_wide: the mat-vec as fp32 v_pk_fma_f32 chains (R, Q register resident) */
    CMPS_WIDE_CHAIN_MFMA_FWD = 2,  /* the forward as MFMA below, the reverse scan as VALU (k_bwd_wide): for A/B measurements */
    CMPS_WIDE_CHAIN_MFMA = 1    /* (a new handle's setting) k_fwd_chain16 / k_bwd_chain16: the correction (Q + s R) u as power-of-two scaled fp16 x 2 split operands on
                                 * v_mfma_f32_16x16x32_f16 (three products, fp32 accumulate: the accuracy class of CMPS_RANK1_F16X2);
                                 * identity part and everything behind the mat-vec in float32 */
};

/* Library version (major * 10000 + minor * 100 + patch). */
int cmps_version(void);

/* Replaces: construction of the model object's constant part, CMPS.__init__ (model.py:9-52).
 * D = hparams.bond_dim. */
int cmps_create(int D, cmps_handle_t* out);
int cmps_destroy(cmps_handle_t h);
const char* cmps_last_error(cmps_handle_t h);
int cmps_set_variant(cmps_handle_t h, int variant);
/* The variant the next launch will use (after AUTO resolution): CMPS_VARIANT_BLOCK, _WAVE, _PAIR, _WAVE32 or _WIDE. */
int cmps_get_variant(cmps_handle_t h);
/* Numerical options of the gradient path; the reference has one arithmetic (TensorFlow float32 kernels behind
 * train.py:89), so every value of every option must stay within the stated float32 tolerance of it.
 * cmps_get_option returns the value, or -1 for an unknown option / null handle. */
int cmps_set_option(cmps_handle_t h, int option, int value);
int cmps_get_option(cmps_handle_t h, int option);
/* With CMPS_OPT_KERNEL_EVENTS on: waits for the recorded events, writes per kernel name (in first-launch order) the summed milliseconds
 * and the launch count since the last call, the names '\n'-joined into `names`; returns the number of distinct kernels (<= cap), -1 on
 * error (option off, bad argument, buffer too small).  Resets the record.  No reference counterpart (see CMPS_OPT_KERNEL_EVENTS). */
int cmps_kernel_times(cmps_handle_t h, char* names, size_t names_bytes, float* ms_sum, int* calls, int cap);

/* Bytes of device workspace the caller must provide for bond dimension D, B clips of T samples.
 * Returns 0 for invalid arguments. */
size_t cmps_workspace_bytes(int D, int B, int T, int flags);

/*
 * Replaces: the parameter-derived constants that the TF graph rebuilds on every session.run --
 * model.py:41-42 (complex R), :52 (freqsc), :304-305 / :321-322 (phases = exp(1j * freqsc * t) for every
 * step, with t accumulated as a sequential float32 sum, model.py:16, 266, 281), :308 (adjoint(R)),
 * :312 (-delta_t * sigma**2 / 2 factor), :221-222 (psi_0).
 * Inputs are the EFFECTIVE parameters (after the rsqrt(reg) scaling and the diagonal removal of
 * model.py:36-42, 49), float32, on the device:
 *   R_re_dev, R_im_dev [D*D] row-major R[i][j];  freqs_dev [D];  psi0_re_dev, psi0_im_dev [D] (normalised).
 * A = model.A (model.py:19), sigma (model.py:21), delta_t (model.py:15), T = samples per clip.
 * Builds, on `stream`, inside `workspace_dev`: R^T, Q = -(delta_t sigma^2 / 2) R^dagger R, the float32
 * time table t_k, and the per-step phase-rotation table.
 * Caching contract: the time table depends only on (T, delta_t).  By default every call rebuilds it (~0.7 ms at T = 16000);
 * a caller that keeps the workspace untouched between calls passes CMPS_WS_REUSE_TABLES and the table is rebuilt only when the
 * workspace address, T or delta_t differ from the handle's previous call.  The workspace is caller-owned memory: after
 * freeing, zeroing, re-obtaining (an allocator may hand back the same address) or sharing it, drop the flag for one call.
 * Everything else in the workspace is rebuilt by every call.
 */
int cmps_set_params(cmps_handle_t h, const float* R_re_dev, const float* R_im_dev,
                    const float* freqs_dev, const float* psi0_re_dev, const float* psi0_im_dev,
                    float A, double sigma, double delta_t, int T, int B_max, int flags,
                    void* workspace_dev, size_t workspace_bytes, void* stream);

/*
 * cmps_set_params with every parameter -- model.A (model.py:19) included -- read from device memory:
 *   params_dev [2 D^2 + 3 D + 1] = R_re [D*D] | R_im [D*D] | freqs [D] | psi0_re [D] | psi0_im [D] | A,
 * the buffer cmps_psi_apply_step writes.  The scan kernels then load A from params_dev + 2 D^2 + 3 D, so a training loop needs
 * no device -> host copy between steps.  Everything else as cmps_set_params.
 */
int cmps_set_params_dev(cmps_handle_t h, const float* params_dev, double sigma, double delta_t, int T, int B_max, int flags,
                        void* workspace_dev, size_t workspace_bytes, void* stream);

/*
 * Replaces: the optimiser half of a training step, which the reference runs inside session.run(train_op) --
 * tf.train.AdamOptimizer(learning_rate).minimize(total_loss) (train.py:88-89; beta1 0.9, beta2 0.999, epsilon 1e-8:
 * logging/graph.pbtxt:32102-32192), i.e. the chain rule from the effective parameters back to the trainable variables
 * (adjoint of model.py:36-42, 49, 221-222), the regularisers total = loss + h_reg sum freqs^2 + r_reg sum |R|^2 of
 * train.py:55-60 (with_reg != 0), the Adam update, and the next step's effective parameters (model.py:36-42, 49, 221-222).
 *   vars_dev, adam_m_dev, adam_v_dev [2 D^2 + 3 D + 1]: A | Rx [D*D] | Ry [D*D] | freqs [D] | psi_x [D] | psi_y [D]  (in / out)
 *   grad_sums_dev [2 D^2 + 3 D + 2]: the buffer of cmps_psi_loss_bwd, summed over all ranks; global_batch = clips it covers.
 *     NULL: no update -- only params_dev is computed from vars_dev (the first step).
 *   lr_t = learning_rate * sqrt(1 - beta2^t) / (1 - beta1^t) for the step count t the caller keeps (train.py:88 global_step).
 *   c_r, c_h: the rsqrt(r_reg), rsqrt(h_reg) factors of model.py:36-39, 49 (1 when R_in / freqs_in were given).
 *   params_dev [2 D^2 + 3 D + 1]: out, the input of cmps_set_params_dev.   losses_dev [2]: out, mean_b loss_b and the total loss
 *   of the gradients consumed.   scratch_dev: cmps_apply_step_scratch_bytes(D) bytes, 8-byte aligned.
 * One small kernel on `stream`; nothing is copied to the host.
 */
size_t cmps_apply_step_scratch_bytes(int D);
int cmps_psi_apply_step(cmps_handle_t h, float* vars_dev, float* adam_m_dev, float* adam_v_dev, const float* grad_sums_dev,
                        double global_batch, double lr_t, double beta1, double beta2, double epsilon, double h_reg, double r_reg,
                        double c_r, double c_h, int with_reg, float* params_dev, float* losses_dev, void* scratch_dev,
                        void* stream);

/*
 * Replaces: PsiCMPS._build_loss_psi (model.py:257-267) = tf.foldl of _psi_and_loss_update
 * (model.py:276-282) over the T-1 increments, i.e. _update_ancilla_psi (:300-317), _inc_loss_psi
 * (:293-294), _expectation (:319-325), _normalize_psi (:327-334) per step.
 * audio_dev [B*T] row-major float32 clips (the `data_iterator` tensor);  loss_dev [B] receives the
 * per-clip loss (the fold carry); PsiCMPS.loss is its mean (model.py:267).
 * save_for_bwd != 0 stashes the per-step un-normalised state in the workspace (needs CMPS_WS_TRAIN).
 */
int cmps_psi_loss_fwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* loss_dev,
                      int save_for_bwd, void* stream);

/*
 * Replaces: the reverse while-loop that tf.train.AdamOptimizer.minimize builds (train.py:89,
 * training_estimators.py:68) for d(sum_b loss_b)/d(effective parameters).
 * Must follow cmps_psi_loss_fwd(..., save_for_bwd=1) on the same audio, B, T and stream.
 * grad_dev [2*D*D + 3*D + 2] receives SUMS over the B clips (divide by the global batch for the mean):
 *   dR_re [D*D], dR_im [D*D] (row-major, dL/dRe R[i][j], dL/dIm R[i][j]), dfreqs [D],
 *   dpsi0_re [D], dpsi0_im [D], dA, sum_b loss_b.
 */
int cmps_psi_loss_bwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* grad_dev,
                      void* stream);

/*
 * Run-time check of the split-operand arithmetic (no reference counterpart: the reference has one float32 arithmetic).
 * cmps_psi_loss_bwd leaves two flag words in the workspace: bit 0 = a gradient sum is Inf / NaN, bit 1 = the loss sum is.
 * This call waits for `stream`, reads them and returns
 *   CMPS_OK             gradient finite, or loss and gradient both non-finite (that propagates, as in the reference);
 *   CMPS_ERR_F16_RANGE  gradient non-finite although the loss is finite: with an F16X2 arithmetic in force an operand piece
 *                       overflowed fp16 (a violated scale bound).  Documented fallback: set CMPS_OPT_RANK1 = BF16X3 and
 *                       CMPS_OPT_WIDE_CHAIN = VALU and repeat cmps_psi_loss_fwd / _bwd (audio_mps_amd.scan.HipScan does so).
 * *sticky_out (may be NULL) receives the OR of the flag words of every cmps_psi_loss_bwd since the previous call of this
 * function, for loops that do not check every step; cmps_psi_apply_step given `status_dev` skips the update of such a step.
 */
int cmps_psi_grad_status(cmps_handle_t h, int* sticky_out, void* stream);

/*
 * Replaces: PsiCMPS._update_ancilla_psi (model.py:300-317), one step in the lab frame.
 * psi_in_dev / psi_out_dev [B*D*2] interleaved (re, im); signal_dev [B]; t = time of the step.
 */
int cmps_psi_update_ancilla(cmps_handle_t h, const float* psi_in_dev, const float* signal_dev,
                            float t, int B, float* psi_out_dev, void* stream);

/*
 * Replaces: PsiCMPS.psi_evolve_with_data (model.py:231-240): the normalised lab-frame state after
 * every step, psi_out_dev [B*(T-1)*D*2] interleaved (re, im), reconstructed from the stash written by
 * cmps_psi_loss_fwd(..., save_for_bwd=1).
 */
int cmps_psi_states(cmps_handle_t h, int B, int T, float* psi_out_dev, void* stream);

/*
 * Replaces: PsiCMPS.sample's tf.scan of _psi_and_sample_update (model.py:242-251, 284-291) for pre-drawn noise
 * (the reference draws it with tf.random_normal([length, n], stddev = sigma * sqrt(temp * delta_t)), model.py:246).
 * noise_dev [n * length] row-major [path][step] (transposed w.r.t. the reference so that a path reads contiguous
 * memory); out_dev [n * length] receives A * (running sum of the increments), i.e. `self.A * transpose(samples)`.
 * Needs cmps_set_params with T >= length + 1 (the per-step time/phase tables); any workspace flags.
 */
int cmps_psi_sample(cmps_handle_t h, const float* noise_dev, int n, int length, float* out_dev, void* stream);

/*
 * Legacy `AudioMPS` arithmetic (the model training_estimators.py:43-45 was written for; its class body is gone from
 * model.py, its training graph survives in logging/graph.pbtxt: psi_0 = e_0, loss += (x - 2 Re<psi|R|psi>)^2 / 2 before
 * the update, psi' = psi + Q psi + dt x R psi, Q = dt (-i H_s - R^T R / 2), graph.pbtxt:10585-14847).
 * R_dev [D*D] real row-major; Q_re_dev, Q_im_dev [D*D] (Q is built by the caller from H and R).
 * cmps_legacy_loss_fwd / _bwd mirror cmps_psi_loss_fwd / _bwd; grad_dev [3*D*D + 1] receives sums over clips of
 *   dQ_re [D*D], dQ_im [D*D], dR (the direct real-R part) [D*D], sum_b loss_b.
 * D <= 32 runs on the wavefront-per-clip kernels of the pure-state path in their legacy mode (same rank-1 arithmetic option,
 * CMPS_OPT_RANK1); 32 < D <= 128 on the wide kernels in their legacy mode (fp32 VALU chains; the H y and gradient GEMMs with two fp16
 * pieces for CMPS_RANK1_F16X2 / DEFAULT, three bf16 pieces otherwise); CMPS_VARIANT_BLOCK runs the general one-workgroup-per-clip
 * kernels at every D (the cross-check implementation).
 */
int cmps_legacy_set_params(cmps_handle_t h, const float* R_dev, const float* Q_re_dev, const float* Q_im_dev,
                           double delta_t, int T, int B_max, int flags, void* workspace_dev, size_t workspace_bytes,
                           void* stream);
int cmps_legacy_loss_fwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* loss_dev, int save_for_bwd,
                         void* stream);
int cmps_legacy_loss_bwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* grad_dev, void* stream);

/*
 * RhoCMPS: the density-matrix scan (model.py:55-203).  All entries need cmps_set_params first (R, freqs, A, sigma,
 * delta_t and the per-step tables are shared with the pure-state path; its psi_0 arguments are ignored here).
 *
 * The reference's rho_0 = W^dagger W / trace (model.py:127-132) has rank `rank` = hparams.initial_rank (default D), and
 * rho -> U rho U^dagger / trace keeps it, so the state is handed over and carried as its `rank` columns:
 *   phi_re_dev, phi_im_dev [rank*D] row-major [a][d],  rho_0 = sum_a phi_a phi_a^dagger,  phi_a = conj(W[a, :]) / sqrt(tr).
 *
 * cmps_rho_workspace_bytes / cmps_rho_set_state: a second caller-owned, 256-B aligned workspace for the columns, their
 *   per-step stash (CMPS_WS_TRAIN: B_max * (T-1) * rank * D' * 8 bytes) and the reduction buffers.  Replaces `_rho_init`
 *   (model.py:119-132) + `tf.stack(batch_size * [self.rho_0])` (:136).  Any rank <= 128 (the reference's default is rank = D,
 *   model.py:62-65): up to rank * D = 5000 the general kernels keep their column arrays in LDS, above that in this workspace
 *   (B_max * 4 * rank * D * 8 more bytes; cmps_rho_sample then needs n <= B_max).
 * cmps_rho_loss_fwd: RhoCMPS._build_loss_rho (model.py:133-144) = tf.foldl of _rho_and_loss_update (:152-158):
 *   _update_ancilla_rho (:172-187), _inc_loss_rho (:166), _expectation (:189-196), _normalize_rho (:198-203).
 *   loss_dev [B] per-clip loss; the caller takes the mean (:144).
 * cmps_rho_loss_bwd: the reverse while-loop TF autodiff builds for that fold.  grad_dev [2*D*D + 3*D + 2 + 2*rank*D]
 *   receives sums over clips:  the cmps_psi_loss_bwd layout (dR_re, dR_im, dfreqs, 2*D unused zeros, dA, sum_b loss_b)
 *   followed by dphi_re [rank*D], dphi_im [rank*D] (cotangents of the columns phi_a).
 * cmps_rho_update_ancilla: RhoCMPS._update_ancilla_rho (model.py:172-187) for arbitrary rho_in_dev [B*D*D*2]
 *   (row-major, interleaved re/im), signal_dev [B], time t; rho_out_dev like rho_in_dev.  Needs no cmps_rho_set_state.
 * cmps_rho_sample: RhoCMPS.sample / rho_evolve_with_sampling / purity (model.py:86-116): tf.scan of
 *   _rho_and_sample_update (:160-167) for pre-drawn noise_dev [n*length] ([path][step]); out_dev [n*length] = A * running
 *   sum.  save_states != 0 keeps the columns of every step (needs a CMPS_WS_TRAIN rho workspace sized for B_max >= n,
 *   T >= length + 1) for cmps_rho_states.  Kernel selection follows cmps_set_variant like the loss entries: D <= 32 and
 *   rank <= 32 run the row-array GEMM kernels (one wavefront per clip / path), CMPS_VARIANT_BLOCK the general ones.
 * cmps_rho_states: lab-frame normalised rho after every step of the last cmps_rho_loss_fwd(save_for_bwd=1) or
 *   cmps_rho_sample(save_states=1): rho_out_dev [B*steps*D*D*2] (rho_evolve_with_data, model.py:76-84 /
 *   rho_evolve_with_sampling, :86-92) and/or purity_out_dev [B*steps] = tr rho^2 (:94-101); either may be NULL.
 */
size_t cmps_rho_workspace_bytes(int D, int rank, int B, int T, int flags);
int cmps_rho_set_state(cmps_handle_t h, const float* phi_re_dev, const float* phi_im_dev, int rank, int T, int B_max,
                       int flags, void* rho_workspace_dev, size_t rho_workspace_bytes, void* stream);
int cmps_rho_loss_fwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* loss_dev, int save_for_bwd,
                      void* stream);
int cmps_rho_loss_bwd(cmps_handle_t h, const float* audio_dev, int B, int T, float* grad_dev, void* stream);
int cmps_rho_update_ancilla(cmps_handle_t h, const float* rho_in_dev, const float* signal_dev, float t, int B,
                            float* rho_out_dev, void* stream);
int cmps_rho_sample(cmps_handle_t h, const float* noise_dev, int n, int length, float* out_dev, int save_states,
                    void* stream);
int cmps_rho_states(cmps_handle_t h, int B, int steps, float* rho_out_dev, float* purity_out_dev, void* stream);

/*
 * Host utility (no GPU involved): CRC-32C (Castagnoli, reflected polynomial 0x82F63B78) of `n` bytes, continuing from
 * `crc_in` (0 to start).  TFRecord framing stores it masked (((crc >> 15) | (crc << 17)) + 0xa282ead8) behind the length
 * and behind the payload of every record; the reference reads those files with tf.data.TFRecordDataset (data.py:29,
 * training_estimators.py:79), which verifies them.  audio_mps_amd/tfrecord.py calls this when verify=True.
 */
unsigned cmps_crc32c(const void* data, size_t n, unsigned crc_in);

#ifdef __cplusplus
}
#endif
#endif /* CMPS_H_ */
