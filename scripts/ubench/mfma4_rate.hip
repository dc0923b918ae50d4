// Microbenchmark: issue rate of v_mfma_f32_4x4x4_16b_bf16 (the pair kernels' mat-vec instruction), one wave per SIMD,
// with 1, 2, 4 and 8 independent accumulators; for comparison v_mfma_f32_32x32x16_bf16 with 4 accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256, 1) void k4(float* out, int iters) {
    s4 a = {1, 2, 3, 4}, b = {(short)threadIdx.x, 1, 2, 3};
    f4 c[8];
    for (int j = 0; j < 8; ++j) c[j] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8 / NACC; ++r)
#pragma unroll
            for (int j = 0; j < NACC; ++j) c[j] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 8; ++j) s += c[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256, 1) void k32(float* out, int iters) {
    s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
// v_mfma_f32_16x16x32_bf16 (the wider candidate for the mat-vecs: 4 of its 16 columns would be used), 4 accumulators
__global__ __launch_bounds__(256, 1) void k16(float* out, int iters) {
    s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f4 c[4];
    for (int j = 0; j < 4; ++j) c[j] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (iters * 8.0);
    printf("%-56s %6.2f ns = %5.1f clk @2.4GHz per instruction per wave\n", name, ns, ns * 2.4);
    (void)hipFree(out);
}
int main() {
    run("mfma 4x4x4_16b bf16, 1 accumulator (dependent chain)", k4<1>);
    run("mfma 4x4x4_16b bf16, 2 accumulators", k4<2>);
    run("mfma 4x4x4_16b bf16, 4 accumulators", k4<4>);
    run("mfma 4x4x4_16b bf16, 8 accumulators", k4<8>);
    run("mfma 32x32x16 bf16, 4 accumulators", k32);
    run("mfma 16x16x32 bf16, 4 accumulators", k16);
    return 0;
}
