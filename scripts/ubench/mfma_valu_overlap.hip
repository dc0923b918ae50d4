// Microbenchmark: how many VALU instructions of the SAME wave hide behind a v_mfma_f32_32x32x16_bf16 (one wave per SIMD)?
// Per iteration: 8 MFMAs on 4 independent accumulators (AGPR or VGPR), each followed by K independent integer / float VALU
// instructions (the instruction mix of the gradient GEMMs' operand build: v_and, v_sub, v_perm), pinned with sched_barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int K>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0;
    float x[8];
    unsigned y[8];
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 0.5f + j; y[j] = threadIdx.x + j; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            c[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < K; ++v) {
                const int j = (r + v) & 7;
                if (v & 1) { y[j] = (y[j] & 0xffff0000u) + 0x10000u; asm volatile("" : "+v"(y[j])); }
                else { x[j] = x[j] - 1.0f; asm volatile("" : "+v"(x[j])); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0];
    for (int j = 0; j < 8; ++j) s += x[j] + (float)y[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename Kn> void run(int K, Kn kern) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("32x32x16 bf16 MFMA + %2d VALU behind each: %6.2f ns per MFMA (%5.1f clk @2.4GHz)\n", K, ms * 1e6 / (iters * 8.0), ms * 1e6 / (iters * 8.0) * 2.4);
    (void)hipFree(out);
}
int main() {
    run(0, k<0>); run(2, k<2>); run(4, k<4>); run(5, k<5>); run(6, k<6>); run(8, k<8>); run(12, k<12>); run(16, k<16>);
    return 0;
}
