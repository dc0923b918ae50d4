// Microbenchmark: 32 v_mfma_f32_4x4x4_16b_bf16 + 32 (or 64) VALU instructions per iteration from a lone wave per SIMD, with the VALU
// placed behind every MFMA, behind every quad, behind every eight, or all behind the 32 MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int GROUP, int VPM>     // GROUP MFMAs, then GROUP * VPM VALU
__global__ __launch_bounds__(256, 1) void k4(float* out, int iters) {
    s4 a = {1, 2, 3, 4}, b = {(short)threadIdx.x, 1, 2, 3};
    f4 c[4];
    float x[8];
    for (int j = 0; j < 4; ++j) c[j] = f4{0, 0, 0, 0};
    for (int j = 0; j < 8; ++j) x[j] = threadIdx.x + j;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 32 / GROUP; ++g) {
#pragma unroll
            for (int r = 0; r < GROUP; ++r) c[r & 3] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c[r & 3], 0, 0, 0);
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
    printf("%-54s %7.1f ns per 32 MFMAs\n", name, ms * 1e6 / iters);
    (void)hipFree(out);
}
int main() {
    run("32 MFMA + 32 VALU: VALU behind every MFMA", k4<1, 1>);
    run("32 MFMA + 32 VALU: 2 + 2", k4<2, 1>);
    run("32 MFMA + 32 VALU: 4 + 4", k4<4, 1>);
    run("32 MFMA + 32 VALU: 8 + 8", k4<8, 1>);
    run("32 MFMA + 32 VALU: 32 + 32", k4<32, 1>);
    run("32 MFMA + 64 VALU: 1 + 2", k4<1, 2>);
    run("32 MFMA + 64 VALU: 4 + 8", k4<4, 2>);
    run("32 MFMA + 64 VALU: 8 + 16", k4<8, 2>);
    run("32 MFMA + 64 VALU: 32 + 64", k4<32, 2>);
    return 0;
}
