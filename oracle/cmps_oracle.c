/*
 * CPU restatement (plain C) of audio-mps's PsiCMPS log-likelihood scan and its gradient.
 *
 * TEST INFRASTRUCTURE ONLY: the checker for tests/, __graft_entry__.smoke() and the cpu_baseline
 * leg of bench.py.  Never linked into, loaded by or called from the product (audio_mps_amd).
 *
 * PARITY UNPINNED: the reference's arithmetic lives in TensorFlow 1.x (requirements.txt:1,
 * unpinned), which cannot be run here, and the reference's tests hold no golden vectors
 * (tests/test_model.py: invariants only).  This file restates model.py's text one clip at a time
 * in the reference's dtypes (float32 / complex64 as explicit re/im pairs, products formed as
 * (ac - bd, ad + bc) with no FMA contraction: build with -ffp-contract=off), with sequential
 * `t += dt` and `loss += ...` accumulation and plain log(1 + z).  It is cross-checked against
 * oracle/cmps_oracle.py (numpy, batched like the TF graph) and its float64 twin.
 *
 * Reference lines followed (paths relative to /root/reference):
 *   increments, fold, carry           model.py:257-267
 *   step order                        model.py:276-282
 *   _update_ancilla_psi               model.py:300-317
 *   _expectation                      model.py:319-325
 *   _inc_loss_psi                     model.py:293-294
 *   _normalize_psi                    model.py:327-334
 * The backward pass has no reference source (tf.train.AdamOptimizer.minimize, train.py:89); it is
 * the op-by-op reverse-mode adjoint of the forward, as in oracle/cmps_oracle.py.
 *
 * One difference from a TF run that is deliberate: per-clip gradient contributions are accumulated
 * in the working precision over time, then summed over clips in double in clip order (TF sums the
 * batch inside its fp32 matmul-gradient kernels in an unknowable order).
 *
 * The file is compiled twice: REAL = float (the stand-in for "TF CPU") and REAL = double (twin).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL float
#define SUFFIX f32
#endif

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)
#define FN(name) CAT(name, SUFFIX)

static inline REAL r_cos(REAL x) { return sizeof(REAL) == 4 ? (REAL)cosf((float)x) : (REAL)cos((double)x); }
static inline REAL r_sin(REAL x) { return sizeof(REAL) == 4 ? (REAL)sinf((float)x) : (REAL)sin((double)x); }
static inline REAL r_log(REAL x) { return sizeof(REAL) == 4 ? (REAL)logf((float)x) : (REAL)log((double)x); }
static inline REAL r_sqrt(REAL x) { return sizeof(REAL) == 4 ? (REAL)sqrtf((float)x) : (REAL)sqrt((double)x); }
static inline REAL r_hypot(REAL x, REAL y) { return sizeof(REAL) == 4 ? (REAL)hypotf((float)x, (float)y) : (REAL)hypot((double)x, (double)y); }

typedef struct {
    int D;
    const REAL *Rre, *Rim, *f;
    REAL A, ccre; /* ccre = (REAL)(-delta_t * sigma^2), formed in double then cast (model.py:312) */
    /* scratch, each D long (re, im) */
    REAL *phr, *phi, *Ur, *Ui, *Vr, *Vi, *Wr, *Wi, *dr, *di, *ppr, *ppi, *Upr, *Upi, *RVr, *RVi;
    /* per-step scalars */
    REAL e, ex_, z, ss, m, inv, s;
} step_t;

/* out[b] = sum_c M[b][c] * in[c]  (einsum 'bc,ac->ab' for one clip a), sequential over c */
static void cmatvec(int D, const REAL* Mre, const REAL* Mim, const REAL* inr, const REAL* ini,
                    REAL* outr, REAL* outi) {
    for (int b = 0; b < D; ++b) {
        REAL ar = 0, ai = 0;
        const REAL *mr = Mre + (size_t)b * D, *mi = Mim + (size_t)b * D;
        for (int c = 0; c < D; ++c) {
            REAL pr = mr[c] * inr[c] - mi[c] * ini[c];
            REAL pi = mr[c] * ini[c] + mi[c] * inr[c];
            ar += pr;
            ai += pi;
        }
        outr[b] = ar;
        outi[b] = ai;
    }
}
/* out[b] = sum_c conj(M[c][b]) * in[c]   (adjoint(R) applied) */
static void cmatvec_adj(int D, const REAL* Mre, const REAL* Mim, const REAL* inr, const REAL* ini,
                        REAL* outr, REAL* outi) {
    for (int b = 0; b < D; ++b) {
        REAL ar = 0, ai = 0;
        for (int c = 0; c < D; ++c) {
            REAL mr = Mre[(size_t)c * D + b], mi = -Mim[(size_t)c * D + b];
            REAL pr = mr * inr[c] - mi * ini[c];
            REAL pi = mr * ini[c] + mi * inr[c];
            ar += pr;
            ai += pi;
        }
        outr[b] = ar;
        outi[b] = ai;
    }
}

/* One forward step from the normalised psi (psr, psi_) with increment x at time t.
 * Fills the scratch of st; the un-normalised psi' is (ppr, ppi). */
static void step_forward(step_t* st, const REAL* psr, const REAL* psi_, REAL x, REAL t) {
    const int D = st->D;
    st->s = x / st->A;                                      /* model.py:303 */
    for (int d = 0; d < D; ++d) {                           /* :304-305 */
        REAL th = st->f[d] * t;
        st->phr[d] = r_cos(th);
        st->phi[d] = r_sin(th);
    }
    for (int d = 0; d < D; ++d) {                           /* :306  psi * conj(phases) */
        REAL cr = st->phr[d], ci = -st->phi[d];
        st->Ur[d] = psr[d] * cr - psi_[d] * ci;
        st->Ui[d] = psr[d] * ci + psi_[d] * cr;
    }
    cmatvec(D, st->Rre, st->Rim, st->Ur, st->Ui, st->Vr, st->Vi);          /* :309 */
    cmatvec_adj(D, st->Rre, st->Rim, st->Vr, st->Vi, st->Wr, st->Wi);      /* :310 */
    for (int d = 0; d < D; ++d) {                           /* :312-313 */
        REAL d1r = (st->ccre * st->Wr[d]) / (REAL)2, d1i = (st->ccre * st->Wi[d]) / (REAL)2;
        st->dr[d] = d1r + st->s * st->Vr[d];
        st->di[d] = d1i + st->s * st->Vi[d];
    }
    for (int d = 0; d < D; ++d) {                           /* :315-317 */
        REAL pr = st->phr[d] * st->dr[d] - st->phi[d] * st->di[d];
        REAL pi = st->phr[d] * st->di[d] + st->phi[d] * st->dr[d];
        st->ppr[d] = psr[d] + pr;
        st->ppi[d] = psi_[d] + pi;
    }
    for (int d = 0; d < D; ++d) {                           /* :322-323 */
        REAL cr = st->phr[d], ci = -st->phi[d];
        st->Upr[d] = st->ppr[d] * cr - st->ppi[d] * ci;
        st->Upi[d] = st->ppr[d] * ci + st->ppi[d] * cr;
    }
    cmatvec(D, st->Rre, st->Rim, st->Upr, st->Upi, st->RVr, st->RVi);      /* :324 */
    REAL exr = 0;
    for (int d = 0; d < D; ++d)                             /* Re(conj(U') * RU') */
        exr += st->Upr[d] * st->RVr[d] + st->Upi[d] * st->RVi[d];
    st->e = (REAL)2 * exr;                                  /* :325 */
    st->ex_ = st->e * x;                                    /* :294 */
    st->z = st->ex_ / st->A;
    REAL ss = 0;
    for (int d = 0; d < D; ++d) {                           /* :331 abs -> square -> sum */
        REAL a = r_hypot(st->ppr[d], st->ppi[d]);
        ss += a * a;
    }
    st->ss = ss;
    st->m = ss > (REAL)1e-12 ? ss : (REAL)1e-12;            /* :332 */
    st->inv = (REAL)1 / r_sqrt(st->m);
}

static REAL* scratch_alloc(step_t* st, int D) {
    REAL* buf = (REAL*)calloc((size_t)16 * D, sizeof(REAL));
    REAL** slots[16] = {&st->phr, &st->phi, &st->Ur, &st->Ui, &st->Vr, &st->Vi, &st->Wr, &st->Wi,
                        &st->dr, &st->di, &st->ppr, &st->ppi, &st->Upr, &st->Upi, &st->RVr, &st->RVi};
    for (int i = 0; i < 16; ++i) *slots[i] = buf + (size_t)i * D;
    return buf;
}

/*
 * data [B,T]; R_re/R_im [D,D] effective R (after model.py:42); freqs [D] effective; psi0 [D] normalised.
 * loss_per_clip [B] out.
 * grad: NULL or [2*D*D + 3*D + 2] out = sums over clips with d/d(loss_b) = 1:
 *        Rbar_re [D*D], Rbar_im [D*D], fbar [D], psi0bar_re [D], psi0bar_im [D], Abar, sum_b loss_b.
 * states: NULL or [B, T-1, D, 2] out, the normalised psi after each step.
 * Returns 0, or -1 on allocation failure.
 */
int FN(cmps_oracle_psi)(int B, int T, int D, const REAL* data, const REAL* R_re, const REAL* R_im,
                        const REAL* freqs, const REAL* psi0_re, const REAL* psi0_im, REAL A,
                        double delta_t, double sigma, REAL* loss_per_clip, REAL* grad, REAL* states,
                        int nthreads) {
    const int N = T - 1;
    const int G = 2 * D * D + 3 * D + 2;
    const REAL dt = (REAL)delta_t;                          /* model.py:16 */
    const REAL ccre = (REAL)(-delta_t * sigma * sigma);
    int fail = 0;
    double* gsum = NULL;
    REAL* gclip = NULL;
    if (grad) {
        gsum = (double*)calloc((size_t)G, sizeof(double));
        gclip = (REAL*)calloc((size_t)B * G, sizeof(REAL));
        if (!gsum || !gclip) { free(gsum); free(gclip); return -1; }
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b) {
        step_t st;
        memset(&st, 0, sizeof st);
        st.D = D; st.Rre = R_re; st.Rim = R_im; st.f = freqs; st.A = A; st.ccre = ccre;
        REAL* buf = scratch_alloc(&st, D);
        REAL* tape = grad ? (REAL*)malloc((size_t)N * 2 * D * sizeof(REAL)) : NULL;  /* psi_k */
        REAL* ttab = grad ? (REAL*)malloc((size_t)(N > 0 ? N : 1) * sizeof(REAL)) : NULL;
        REAL* cur = (REAL*)malloc((size_t)2 * D * sizeof(REAL));
        REAL* wk = (REAL*)calloc((size_t)12 * D, sizeof(REAL));
        if (!buf || !cur || !wk || (grad && (!tape || !ttab))) {
#pragma omp atomic write
            fail = 1;
            free(buf); free(tape); free(ttab); free(cur); free(wk);
            continue;
        }
        REAL *psr = cur, *psi_ = cur + D;
        for (int d = 0; d < D; ++d) { psr[d] = psi0_re[d]; psi_[d] = psi0_im[d]; }   /* :260 */
        const REAL* xrow = data + (size_t)b * T;
        REAL loss = 0, t = 0;                                /* :266 */
        for (int k = 0; k < N; ++k) {                        /* :265 foldl */
            if (tape) {
                memcpy(tape + (size_t)k * 2 * D, psr, D * sizeof(REAL));
                memcpy(tape + (size_t)k * 2 * D + D, psi_, D * sizeof(REAL));
                ttab[k] = t;
            }
            REAL x = xrow[k + 1] - xrow[k];                  /* :263 */
            step_forward(&st, psr, psi_, x, t);              /* :278 */
            loss += -r_log((REAL)1 + st.z);                  /* :279, :294 */
            for (int d = 0; d < D; ++d) {                    /* :280, :333-334 */
                psr[d] = st.ppr[d] * st.inv;
                psi_[d] = st.ppi[d] * st.inv;
            }
            t += dt;                                         /* :281 */
            if (states) {
                REAL* o = states + (((size_t)b * N + k) * D) * 2;
                for (int d = 0; d < D; ++d) { o[2 * d] = psr[d]; o[2 * d + 1] = psi_[d]; }
            }
        }
        loss_per_clip[b] = loss;
        if (grad) {
            REAL* gc = gclip + (size_t)b * G;
            REAL *Rbr = gc, *Rbi = gc + D * D, *fb = gc + 2 * D * D, *p0r = fb + D, *p0i = p0r + D;
            REAL Abar = 0;
            REAL *gr = wk, *gi = wk + D, *ppbr = wk + 2 * D, *ppbi = wk + 3 * D, *Upbr = wk + 4 * D,
                 *Upbi = wk + 5 * D, *phbr = wk + 6 * D, *phbi = wk + 7 * D, *dbr = wk + 8 * D,
                 *dbi = wk + 9 * D, *Vbr = wk + 10 * D, *Vbi = wk + 11 * D;
            REAL* wk2 = (REAL*)calloc((size_t)6 * D, sizeof(REAL));
            REAL *Wbr = wk2, *Wbi = wk2 + D, *Ubr = wk2 + 2 * D, *Ubi = wk2 + 3 * D, *tr = wk2 + 4 * D,
                 *ti = wk2 + 5 * D;
            for (int d = 0; d < D; ++d) gr[d] = gi[d] = 0;
            for (int k = N - 1; k >= 0; --k) {
                const REAL* pkr = tape + (size_t)k * 2 * D;
                const REAL* pki = pkr + D;
                REAL x = xrow[k + 1] - xrow[k];
                REAL tk = ttab[k];
                step_forward(&st, pkr, pki, x, tk);
                /* normalise: psi_next = psi' * inv */
                REAL inv_bar = 0;
                for (int d = 0; d < D; ++d) {
                    ppbr[d] = gr[d] * st.inv;
                    ppbi[d] = gi[d] * st.inv;
                    inv_bar += gr[d] * st.ppr[d] + gi[d] * st.ppi[d];
                }
                REAL m_bar = inv_bar * ((REAL)-0.5 * st.inv / st.m);
                REAL ss_bar = st.ss > (REAL)1e-12 ? m_bar : (REAL)0;
                for (int d = 0; d < D; ++d) {
                    ppbr[d] += (REAL)2 * ss_bar * st.ppr[d];
                    ppbi[d] += (REAL)2 * ss_bar * st.ppi[d];
                }
                /* loss increment */
                REAL z_bar = (REAL)-1 / ((REAL)1 + st.z);
                Abar += z_bar * (-st.ex_ / (st.A * st.A));
                REAL e_bar = z_bar * x / st.A;
                REAL ex_bar = (REAL)2 * e_bar;
                for (int d = 0; d < D; ++d) {       /* Up_bar = ex_bar*RVp ; RVp_bar = ex_bar*Up */
                    Upbr[d] = ex_bar * st.RVr[d];
                    Upbi[d] = ex_bar * st.RVi[d];
                    tr[d] = ex_bar * st.Upr[d];
                    ti[d] = ex_bar * st.Upi[d];
                }
                /* Up_bar += RVp_bar @ conj(R):  Up_bar[c] += sum_b RVp_bar[b] * conj(R[b][c]);
                 * Rbar[b][c] += conj(Up[c]) * RVp_bar[b] */
                for (int bb = 0; bb < D; ++bb) {
                    for (int c = 0; c < D; ++c) {
                        REAL rr = R_re[(size_t)bb * D + c], ri = -R_im[(size_t)bb * D + c];
                        Upbr[c] += tr[bb] * rr - ti[bb] * ri;
                        Upbi[c] += tr[bb] * ri + ti[bb] * rr;
                        REAL ur = st.Upr[c], ui = -st.Upi[c];
                        Rbr[(size_t)bb * D + c] += ur * tr[bb] - ui * ti[bb];
                        Rbi[(size_t)bb * D + c] += ur * ti[bb] + ui * tr[bb];
                    }
                }
                /* psi'_bar += Up_bar * ph ; ph_bar = conj(Up_bar * conj(psi')) */
                for (int d = 0; d < D; ++d) {
                    ppbr[d] += Upbr[d] * st.phr[d] - Upbi[d] * st.phi[d];
                    ppbi[d] += Upbr[d] * st.phi[d] + Upbi[d] * st.phr[d];
                    REAL qr = Upbr[d] * st.ppr[d] + Upbi[d] * st.ppi[d];
                    REAL qi = Upbi[d] * st.ppr[d] - Upbr[d] * st.ppi[d];
                    phbr[d] = qr;
                    phbi[d] = -qi;
                }
                /* psi' = psi_k + ph * delta */
                for (int d = 0; d < D; ++d) {
                    gr[d] = ppbr[d];
                    gi[d] = ppbi[d];
                    /* ph_bar += psi'_bar * conj(delta) */
                    phbr[d] += ppbr[d] * st.dr[d] + ppbi[d] * st.di[d];
                    phbi[d] += ppbi[d] * st.dr[d] - ppbr[d] * st.di[d];
                    /* delta_bar = psi'_bar * conj(ph) */
                    dbr[d] = ppbr[d] * st.phr[d] + ppbi[d] * st.phi[d];
                    dbi[d] = ppbi[d] * st.phr[d] - ppbr[d] * st.phi[d];
                }
                /* delta = cc*W/2 + s*V */
                REAL s_bar = 0;
                for (int d = 0; d < D; ++d) {
                    s_bar += dbr[d] * st.Vr[d] + dbi[d] * st.Vi[d];
                    Vbr[d] = dbr[d] * st.s;
                    Vbi[d] = dbi[d] * st.s;
                    Wbr[d] = dbr[d] * (st.ccre / (REAL)2);
                    Wbi[d] = dbi[d] * (st.ccre / (REAL)2);
                }
                /* W[b] = sum_c conj(R[c][b]) V[c]:  V_bar[c] += sum_b W_bar[b] * R[c][b];
                 * Rbar[c][b] += conj(conj(V[c]) * W_bar[b]) = V[c] * conj(W_bar[b]) */
                for (int c = 0; c < D; ++c) {
                    for (int bb = 0; bb < D; ++bb) {
                        REAL rr = R_re[(size_t)c * D + bb], ri = R_im[(size_t)c * D + bb];
                        Vbr[c] += Wbr[bb] * rr - Wbi[bb] * ri;
                        Vbi[c] += Wbr[bb] * ri + Wbi[bb] * rr;
                        Rbr[(size_t)c * D + bb] += st.Vr[c] * Wbr[bb] + st.Vi[c] * Wbi[bb];
                        Rbi[(size_t)c * D + bb] += st.Vi[c] * Wbr[bb] - st.Vr[c] * Wbi[bb];
                    }
                }
                /* V[b] = sum_c R[b][c] U[c]:  U_bar[c] = sum_b V_bar[b] conj(R[b][c]);
                 * Rbar[b][c] += conj(U[c]) * V_bar[b] */
                for (int c = 0; c < D; ++c) Ubr[c] = Ubi[c] = 0;
                for (int bb = 0; bb < D; ++bb) {
                    for (int c = 0; c < D; ++c) {
                        REAL rr = R_re[(size_t)bb * D + c], ri = -R_im[(size_t)bb * D + c];
                        Ubr[c] += Vbr[bb] * rr - Vbi[bb] * ri;
                        Ubi[c] += Vbr[bb] * ri + Vbi[bb] * rr;
                        REAL ur = st.Ur[c], ui = -st.Ui[c];
                        Rbr[(size_t)bb * D + c] += ur * Vbr[bb] - ui * Vbi[bb];
                        Rbi[(size_t)bb * D + c] += ur * Vbi[bb] + ui * Vbr[bb];
                    }
                }
                /* U = psi_k * conj(ph):  g += U_bar * ph ; ph_bar += conj(U_bar * conj(psi_k)) */
                for (int d = 0; d < D; ++d) {
                    gr[d] += Ubr[d] * st.phr[d] - Ubi[d] * st.phi[d];
                    gi[d] += Ubr[d] * st.phi[d] + Ubi[d] * st.phr[d];
                    REAL qr = Ubr[d] * pkr[d] + Ubi[d] * pki[d];
                    REAL qi = Ubi[d] * pkr[d] - Ubr[d] * pki[d];
                    phbr[d] += qr;
                    phbi[d] += -qi;
                }
                Abar += s_bar * (-x / (st.A * st.A));
                /* ph = exp(1j f t): w_bar = ph_bar * conj(ph); fbar += t * Im(w_bar) */
                for (int d = 0; d < D; ++d) {
                    REAL wi = phbi[d] * st.phr[d] - phbr[d] * st.phi[d];
                    fb[d] += tk * wi;
                }
            }
            for (int d = 0; d < D; ++d) { p0r[d] = gr[d]; p0i[d] = gi[d]; }
            gc[2 * D * D + 3 * D] = Abar;
            gc[2 * D * D + 3 * D + 1] = loss;
            free(wk2);
        }
        free(buf); free(tape); free(ttab); free(cur); free(wk);
    }
    if (grad) {
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < G; ++i) gsum[i] += (double)gclip[(size_t)b * G + i];
        for (int i = 0; i < G; ++i) grad[i] = (REAL)gsum[i];
        free(gsum);
        free(gclip);
    }
    return fail ? -1 : 0;
}
