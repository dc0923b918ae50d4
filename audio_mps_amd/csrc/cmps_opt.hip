// Device-resident optimiser step: the chain rule from the reduced effective-parameter gradient sums back to
// the raw variables -- adjoint of model.py:36-42 (rsqrt(reg) scaling + the row-broadcast diagonal removal), :49, :221-222 --
// the regularisers of train.py:55-60, tf.train.AdamOptimizer (train.py:89; beta1 0.9, beta2 0.999, eps 1e-8 from
// logging/graph.pbtxt:32102-32192) and the next step's effective parameters (model.py:36-42, 49, 221-222), all in ONE
// workgroup on the caller's stream: a training step then has no device -> host synchronisation and no host arithmetic.
// The host implementation (audio_mps_amd/model.py::chain_rule, train.py::AdamOptimizer) stays the tested reference for it:
// the chain rule is evaluated in double like numpy's, Adam in float32 with numpy's rounding points (no contraction).
//
//   vars / adam_m / adam_v : [A | Rx D^2 | Ry D^2 | freqs D | psi_x D | psi_y D]           (2 D^2 + 3 D + 1 floats)
//   grad_sums              : what cmps_psi_loss_bwd writes (sums over clips)                 (2 D^2 + 3 D + 2 floats)
//   params_out             : [R_re D^2 | R_im D^2 | freqs D | psi0_re D | psi0_im D | A]    (input of cmps_set_params_dev)
//   losses_out             : [mean_b loss_b, total loss (with regularisers)]
#include "cmps_internal.h"

namespace cmps {

namespace {

constexpr int ONT = 1024;

__device__ __forceinline__ double block_sum_d(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < ONT / 64; ++w) s += red[w];
    return s;
}

// effective R of model.py:36-42 from the raw variables, in the host's float32 rounding: Z = fl(c_r Rx) + i fl(c_r Ry), R[i][j] = Z[i][j] - Z[j][j]
__device__ __forceinline__ float2 eff_R(const float* Rx, const float* Ry, int D, int i, int j, float c_r, bool scaled) {
    const float zx = scaled ? __fmul_rn(c_r, Rx[i * D + j]) : Rx[i * D + j], zy = scaled ? __fmul_rn(c_r, Ry[i * D + j]) : Ry[i * D + j];
    const float dx = scaled ? __fmul_rn(c_r, Rx[j * D + j]) : Rx[j * D + j], dy = scaled ? __fmul_rn(c_r, Ry[j * D + j]) : Ry[j * D + j];
    return make_float2(__fsub_rn(zx, dx), __fsub_rn(zy, dy));
}

// tf.train.AdamOptimizer with numpy's float32 rounding points (audio_mps_amd/train.py::AdamOptimizer.apply_gradients)
__device__ __forceinline__ void adam(float* var, float* m, float* v, int idx, float g, float lr_t, float b1, float omb1, float b2,
                                     float omb2, float eps) {
    const float mn = __fadd_rn(__fmul_rn(b1, m[idx]), __fmul_rn(omb1, g));
    const float vn = __fadd_rn(__fmul_rn(b2, v[idx]), __fmul_rn(__fmul_rn(omb2, g), g));
    m[idx] = mn;
    v[idx] = vn;
    var[idx] = __fsub_rn(var[idx], __fdiv_rn(__fmul_rn(lr_t, mn), __fadd_rn(__fsqrt_rn(vn), eps)));
}

struct OptArgs {
    int D, apply;
    double inv_batch, h_reg, r_reg;
    float c_r, c_h, lr_t, b1, omb1, b2, omb2, eps;
    int scaled_r, scaled_h, with_reg;
};

__global__ __launch_bounds__(ONT) void k_apply_step(OptArgs a, float* __restrict__ vars, float* __restrict__ am, float* __restrict__ av,
                                                    const float* __restrict__ gs, float* __restrict__ params,
                                                    float* __restrict__ losses, double* __restrict__ colsum) {
    __shared__ double red[ONT / 64];
    const int D = a.D, DD = D * D, t = threadIdx.x;
    float* Rx = vars + 1;
    float* Ry = Rx + DD;
    float* fr = Ry + DD;
    float* px = fr + D;
    float* py = px + D;
    // A gradient with Inf / NaN in it next to a FINITE loss sum is the signature of an fp16-split operand that left its scaled range
    // (include/cmps.h: CMPS_ERR_F16_RANGE): such a step is skipped -- variables and Adam slots stay as they are, losses[1] = NaN marks
    // it -- instead of poisoning every variable.  The decision reads the all-reduced buffer, so every rank takes the same one.
    bool skip = false;
    if (a.apply) {
        int bad = 0;
        for (int idx = t; idx < 2 * DD + 3 * D + 1; idx += ONT) bad |= !isfinite(gs[idx]);
        skip = __syncthreads_or(bad) != 0 && isfinite(gs[2 * DD + 3 * D + 1]);
        if (skip && t == 0) {
            losses[0] = (float)((double)gs[2 * DD + 3 * D + 1] * a.inv_batch);
            losses[1] = __builtin_nanf("");
        }
    }
    if (a.apply && !skip) {
        // ---- regulariser terms on the CURRENT effective parameters (train.py:55-60) ----
        double sf = 0.0, sr = 0.0;
        for (int d = t; d < D; d += ONT) {
            const double f = a.scaled_h ? (double)__fmul_rn(a.c_h, fr[d]) : (double)fr[d];
            sf += f * f;
        }
        for (int idx = t; idx < DD; idx += ONT) {
            const float2 r = eff_R(Rx, Ry, D, idx / D, idx % D, a.c_r, a.scaled_r);
            sr += (double)r.x * r.x + (double)r.y * r.y;
        }
        sf = block_sum_d(sf, red);
        sr = block_sum_d(sr, red);
        // ---- column sums of Rbar (adjoint of the row-broadcast diagonal removal, model.py:42) ----
        const double wr = a.with_reg ? 2.0 * a.r_reg : 0.0;
        for (int j = t; j < D; j += ONT) {
            double cx = 0.0, cy = 0.0;
            for (int i = 0; i < D; ++i) {
                const float2 r = eff_R(Rx, Ry, D, i, j, a.c_r, a.scaled_r);
                cx += (double)gs[i * D + j] * a.inv_batch + wr * r.x;
                cy += (double)gs[DD + i * D + j] * a.inv_batch + wr * r.y;
            }
            colsum[2 * j] = cx;
            colsum[2 * j + 1] = cy;
        }
        // ---- psi_0 adjoint (model.py:221-222): p0bar -> (psi_x, psi_y) ----
        double ss = 0.0, ib = 0.0;
        for (int d = t; d < D; d += ONT) {
            const double x = px[d], y = py[d];
            ss += x * x + y * y;
            ib += (double)gs[2 * DD + D + d] * a.inv_batch * x + (double)gs[2 * DD + 2 * D + d] * a.inv_batch * y;   // Re(conj(p0bar) p)
        }
        ss = block_sum_d(ss, red);
        ib = block_sum_d(ib, red);
        __syncthreads();                                           // colsum visible
        const double mm = ss > 1e-12 ? ss : 1e-12, inv = 1.0 / sqrt(mm);
        const double pc = ss > 1e-12 ? 2.0 * (ib * (-0.5 * inv / mm)) : 0.0;
        if (t == 0) {
            const double loss = (double)gs[2 * DD + 3 * D + 1] * a.inv_batch;
            losses[0] = (float)loss;
            losses[1] = (float)(a.with_reg ? loss + a.h_reg * sf + a.r_reg * sr : loss);
        }
        // ---- gradients w.r.t. the raw variables + Adam ----
        // R's diagonal feeds a whole column of the effective R: every gradient is formed from the OLD variables (into the
        // scratch buffer) before any variable is updated
        float* gxy = reinterpret_cast<float*>(colsum + 2 * D);
        for (int idx = t; idx < DD; idx += ONT) {
            const int i = idx / D, j = idx % D;
            const float2 r = eff_R(Rx, Ry, D, i, j, a.c_r, a.scaled_r);
            double zx = (double)gs[idx] * a.inv_batch + wr * r.x, zy = (double)gs[DD + idx] * a.inv_batch + wr * r.y;
            if (i == j) { zx -= colsum[2 * j]; zy -= colsum[2 * j + 1]; }
            gxy[2 * idx] = (float)((a.scaled_r ? (double)a.c_r : 1.0) * zx);
            gxy[2 * idx + 1] = (float)((a.scaled_r ? (double)a.c_r : 1.0) * zy);
        }
        __syncthreads();
        for (int idx = t; idx < DD; idx += ONT) {
            adam(vars, am, av, 1 + idx, gxy[2 * idx], a.lr_t, a.b1, a.omb1, a.b2, a.omb2, a.eps);
            adam(vars, am, av, 1 + DD + idx, gxy[2 * idx + 1], a.lr_t, a.b1, a.omb1, a.b2, a.omb2, a.eps);
        }
        for (int d = t; d < D; d += ONT) {
            const double f = a.scaled_h ? (double)__fmul_rn(a.c_h, fr[d]) : (double)fr[d];
            const double fb = (double)gs[2 * DD + d] * a.inv_batch + (a.with_reg ? 2.0 * a.h_reg * f : 0.0);
            const float gf = (float)((a.scaled_h ? (double)a.c_h : 1.0) * fb);
            const double bx = (double)gs[2 * DD + D + d] * a.inv_batch, by = (double)gs[2 * DD + 2 * D + d] * a.inv_batch;
            const float gpx = (float)(bx * inv + pc * (double)px[d]), gpy = (float)(by * inv + pc * (double)py[d]);
            adam(vars, am, av, 1 + 2 * DD + d, gf, a.lr_t, a.b1, a.omb1, a.b2, a.omb2, a.eps);
            adam(vars, am, av, 1 + 2 * DD + D + d, gpx, a.lr_t, a.b1, a.omb1, a.b2, a.omb2, a.eps);
            adam(vars, am, av, 1 + 2 * DD + 2 * D + d, gpy, a.lr_t, a.b1, a.omb1, a.b2, a.omb2, a.eps);
        }
        if (t == 0) adam(vars, am, av, 0, (float)((double)gs[2 * DD + 3 * D] * a.inv_batch), a.lr_t, a.b1, a.omb1, a.b2, a.omb2, a.eps);
        __syncthreads();
    }
    // ---- effective parameters of the (updated) variables: model.py:36-42, 49, 221-222 ----
    for (int idx = t; idx < DD; idx += ONT) {
        const float2 r = eff_R(Rx, Ry, D, idx / D, idx % D, a.c_r, a.scaled_r);
        params[idx] = r.x;
        params[DD + idx] = r.y;
    }
    float ssf = 0.f;
    for (int d = t; d < D; d += ONT) {
        params[2 * DD + d] = a.scaled_h ? __fmul_rn(a.c_h, fr[d]) : fr[d];
        const float ab = hypotf(px[d], py[d]);                    // tf.abs of a complex64, then tf.square (model.py:331)
        ssf += ab * ab;
    }
    const float ss32 = (float)block_sum_d((double)ssf, red);
    const float invn = __fdiv_rn(1.0f, __fsqrt_rn(fmaxf(ss32, 1e-12f)));
    for (int d = t; d < D; d += ONT) {
        params[2 * DD + D + d] = __fmul_rn(px[d], invn);
        params[2 * DD + 2 * D + d] = __fmul_rn(py[d], invn);
    }
    if (t == 0) params[2 * DD + 3 * D] = vars[0];
}

}  // namespace

size_t apply_step_scratch_bytes(int D) { return (size_t)2 * D * sizeof(double) + (size_t)2 * D * D * sizeof(float); }

hipError_t launch_apply_step(int D, bool apply, double inv_batch, double lr_t, double beta1, double beta2, double eps, double h_reg,
                             double r_reg, double c_r, double c_h, bool with_reg, float* vars, float* am, float* av,
                             const float* grad_sums, float* params_out, float* losses_out, double* scratch, hipStream_t s) {
    OptArgs a{};
    a.D = D; a.apply = apply ? 1 : 0;
    a.inv_batch = inv_batch; a.h_reg = h_reg; a.r_reg = r_reg;
    a.c_r = (float)c_r; a.c_h = (float)c_h;
    a.scaled_r = c_r != 1.0; a.scaled_h = c_h != 1.0; a.with_reg = with_reg ? 1 : 0;
    a.lr_t = (float)lr_t;
    a.b1 = (float)beta1; a.omb1 = (float)(1.0 - beta1);           // numpy: a python float times a float32 array rounds the scalar to float32
    a.b2 = (float)beta2; a.omb2 = (float)(1.0 - beta2);
    a.eps = (float)eps;
    hipLaunchKernelGGL(k_apply_step, dim3(1), dim3(ONT), 0, s, a, vars, am, av, grad_sums, params_out, losses_out, scratch);
    return hipGetLastError();
}

}  // namespace cmps
