// Microbenchmark: v_mfma_f32_32x32x16_bf16 from a lone wave per SIMD with V VALU instructions per MFMA, placed behind every MFMA,
// behind every pair, or behind every four.  ns per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int GROUP, int VPM>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f16 c[4];
    float x[8];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0;
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 8 / GROUP; ++g) {
#pragma unroll
            for (int r = 0; r < GROUP; ++r) c[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < GROUP * VPM; ++v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[v & 7]) : "v"(x[(v + 3) & 7]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0];
    for (int j = 0; j < 8; ++j) s += x[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 10000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %6.2f ns per MFMA\n", name, ms * 1e6 / (iters * 8.0));
    (void)hipFree(out);
}
int main() {
    run("3 VALU per MFMA: 1 + 3", k<1, 3>); run("3 VALU per MFMA: 2 + 6", k<2, 3>); run("3 VALU per MFMA: 4 + 12", k<4, 3>);
    run("4 VALU per MFMA: 1 + 4", k<1, 4>); run("4 VALU per MFMA: 2 + 8", k<2, 4>); run("4 VALU per MFMA: 4 + 16", k<4, 4>);
    run("5 VALU per MFMA: 1 + 5", k<1, 5>); run("5 VALU per MFMA: 2 + 10", k<2, 5>); run("5 VALU per MFMA: 4 + 20", k<4, 5>);
    run("6 VALU per MFMA: 1 + 6", k<1, 6>); run("6 VALU per MFMA: 2 + 12", k<2, 6>); run("6 VALU per MFMA: 4 + 24", k<4, 6>);
    return 0;
}
