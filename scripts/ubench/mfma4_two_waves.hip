// Microbenchmark: v_mfma_f32_4x4x4_16b_bf16 issued by ONE or TWO waves per SIMD (256 / 512 threads per workgroup, one workgroup per
// CU), four accumulators in rotation; with 0 or 2 VALU instructions behind each MFMA.  Time per MFMA per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NT, int KV>
__global__ __launch_bounds__(NT, 1) void k4(float* out, int iters) {
    s4 a = {1, 2, 3, 4}, b = {(short)threadIdx.x, 1, 2, 3};
    f4 c[4];
    float x[4];
    for (int j = 0; j < 4; ++j) { c[j] = f4{0, 0, 0, 0}; x[j] = threadIdx.x + j; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            c[r & 3] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(a, b, c[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < KV; ++v) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[(r + v) & 3]) : "v"(x[(r + v + 1) & 3]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0] + x[j];
    out[blockIdx.x * NT + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, int nt) {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = ms * 1e6 / (iters * 8.0 * (nt / 256));
    printf("%-60s %6.2f ns per MFMA per SIMD\n", name, per_simd);
    (void)hipFree(out);
}
int main() {
    run("4x4x4 bf16, one wave per SIMD", k4<256, 0>, 256);
    run("4x4x4 bf16, two waves per SIMD", k4<512, 0>, 512);
    run("4x4x4 bf16 + 2 VALU each, one wave per SIMD", k4<256, 2>, 256);
    run("4x4x4 bf16 + 2 VALU each, two waves per SIMD", k4<512, 2>, 512);
    run("4x4x4 bf16 + 1 VALU each, one wave per SIMD", k4<256, 1>, 256);
    run("4x4x4 bf16 + 1 VALU each, two waves per SIMD", k4<512, 1>, 512);
    return 0;
}
