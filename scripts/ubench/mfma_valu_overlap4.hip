// Microbenchmark 4: four VALU instructions behind each v_mfma_f32_32x32x16_bf16 (one wave per SIMD) -- independent, a dependent chain
// (the bf16 split: and -> sub -> and -> sub), or two interleaved chains.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0;
    float x[8], y[8], z[8];
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 0.5f + j; y[j] = j; z[j] = 1; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            c[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            const int j = r, j2 = (r + 1) & 7;
            if (MODE == 0) {          // independent
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j]) : "v"(x[j]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j]) : "v"(x[j2]), "v"(x[j]));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j2]) : "v"(x[j2]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j2]) : "v"(x[j]), "v"(x[j2]));
            } else if (MODE == 1) {   // one dependent chain
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j]) : "v"(x[j]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j]) : "v"(x[j]), "v"(y[j]));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j2]) : "v"(z[j]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j2]) : "v"(z[j]), "v"(y[j2]));
            } else if (MODE == 2) {   // two chains of two, interleaved
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j]) : "v"(x[j]));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j2]) : "v"(x[j2]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j]) : "v"(x[j]), "v"(y[j]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j2]) : "v"(x[j2]), "v"(y[j2]));
            } else if (MODE == 3) {   // chain with one independent instruction between dependents
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j]) : "v"(x[j]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j]) : "v"(x[j]), "v"(y[j]));
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j2]) : "v"(x[j2]));
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(z[j2]) : "v"(x[j2]), "v"(y[j2]));
            } else if (MODE == 4) {   // three v_perm reading fresh results
                asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(y[j]) : "v"(x[j]));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(z[j]) : "v"(y[j]), "v"(x[j2]), "s"(0x07060302u));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(z[j2]) : "v"(y[j]), "v"(x[j]), "s"(0x07060302u));
                asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(y[j2]) : "v"(z[j2]), "v"(z[j]), "s"(0x07060302u));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0];
    for (int j = 0; j < 8; ++j) s += x[j] + y[j] + z[j];
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
    printf("%-52s : %6.2f ns per MFMA\n", name, ms * 1e6 / (iters * 8.0));
}
int main() {
    (void)hipMalloc(&g_out, 256 * 256 * 4);
    run("4 independent VALU", k<0>);
    run("4 VALU, one dependent chain (and-sub-and-sub)", k<1>);
    run("4 VALU, two chains of two interleaved", k<2>);
    run("4 VALU, two chains of two back to back", k<3>);
    run("and + 3 v_perm on fresh results", k<4>);
    run("4 independent VALU (again)", k<0>);
    return 0;
}
