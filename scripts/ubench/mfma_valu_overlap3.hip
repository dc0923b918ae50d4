// Microbenchmark 3: a GEMM-like stream of one wave per SIMD (4 waves per CU): every v_mfma_f32_32x32x16_bf16 takes its A operand from
// LDS (ds_read_b128 issued RD MFMAs ahead, counted lgkmcnt wait), with K VALU instructions and, every fourth MFMA, W ds_write_b64
// behind it.  Which ingredient keeps the VALU work from hiding behind the matrix pipe?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int K, int W, bool READS>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ f4 lds[4096];
    s8 b = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0;
    float x[8];
    unsigned y[8];
    f4 q[4];
    float2 p = make_float2(1.f, 2.f);
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 0.5f + j; y[j] = threadIdx.x + j; }
    for (int j = 0; j < 4; ++j) q[j] = f4{1, 2, 3, 4};
    for (int j = threadIdx.x; j < 4096; j += 256) lds[j] = f4{1, 2, 3, 4};
    __syncthreads();
    const unsigned la = threadIdx.x * 16, lw = 32768 + threadIdx.x * 8;
    const unsigned sel = 0x07060302u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (READS) asm volatile("s_waitcnt lgkmcnt(%0)" : : "n"(3 + (W > 0 ? W : 0)));   // the oldest read has landed (at most 3 reads + the stores behind it)
            s8 a;
            __builtin_memcpy(&a, &q[r & 3], 16);
            asm volatile("" : "+v"(a));
            c[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (READS) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[r & 3]) : "v"(la), "n"((r & 3) * 4096));
#pragma unroll
            for (int v = 0; v < K; ++v) {
                const int j = (r + v) & 7;
                const int kind = v % 4;
                if (kind == 0) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(y[j]));
                if (kind == 1) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(y[j]) : "v"(y[(j + 1) & 7]), "s"(sel));
                if (kind == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(x[(j + 3) & 7]));
                if (kind == 3) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(x[(j + 1) & 7]));
            }
            if ((r & 3) == 3)
#pragma unroll
                for (int v = 0; v < W; ++v) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(lw), "v"(p), "n"(v * 2048));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0] + q[j][0];
    for (int j = 0; j < 8; ++j) s += x[j] + (float)y[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static float* g_out;
template <typename Kn> void run(const char* name, Kn kern) {
    const int iters = 10000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, g_out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, g_out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s : %6.2f ns per MFMA\n", name, ms * 1e6 / (iters * 8.0));
}
int main() {
    (void)hipMalloc(&g_out, 256 * 256 * 4);
    run("MFMA only (operands in registers)", k<0, 0, false>);
    run("MFMA + 1 ds_read_b128 each", k<0, 0, true>);
    run("MFMA + read + 2 VALU", k<2, 0, true>);
    run("MFMA + read + 3 VALU", k<3, 0, true>);
    run("MFMA + read + 4 VALU", k<4, 0, true>);
    run("MFMA + read + 5 VALU", k<5, 0, true>);
    run("MFMA + read + 4 VALU + 1 ds_write_b64 / 4", k<4, 1, true>);
    run("MFMA + read + 4 VALU + 2 ds_write_b64 / 4", k<4, 2, true>);
    run("MFMA + read + 4 VALU + 4 ds_write_b64 / 4", k<4, 4, true>);
    run("MFMA + read + 3 VALU + 1 ds_write_b64 / 4", k<3, 1, true>);
    run("MFMA + 4 VALU + 1 ds_write_b64 / 4 (no reads)", k<4, 1, false>);
    return 0;
}
