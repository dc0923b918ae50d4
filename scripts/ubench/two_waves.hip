// Microbenchmark: do two waves on one SIMD issue at twice the rate of a lone wave?  And what does a
// workgroup-wide s_barrier per iteration cost?  (512-thread workgroups = 8 waves per CU = 2 per SIMD.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define REP16(x) x x x x x x x x x x x x x x x x
// MODE 0: all waves 64 pk_fma per iter.  MODE 1: waves 0-3 64 pk_fma, waves 4-7 6 MFMA + 16 pk_fma.
// MODE 2: as 1 plus one s_barrier per iter.  MODE 3: as 0 plus s_barrier.  MODE 4: 64 pk + 40 v_add mix, all waves
template <int MODE, int NT>
__global__ __launch_bounds__(NT, 1) void k(float* out, long long* cyc, int iters) {
    v2f a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f}, m = {1.0001f, 0.9999f}, b = {0.5f, 0.25f};
    v16f acc0 = {}, acc1 = {};
    float ma = threadIdx.x * 1e-3f, mb = 1.f;
    const int w = threadIdx.x >> 6;
    const bool roleB = (MODE == 1 || MODE == 2) && w >= 4;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (!roleB) {
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(b));)
        } else {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                         "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                         "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3\n"
                         "v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(b));
        }
        if (MODE == 2 || MODE == 3) __builtin_amdgcn_s_barrier();
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y;
    for (int q = 0; q < 16; ++q) r += acc0[q] + acc1[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}
template <int MODE, int NT> void run(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * NT * sizeof(float)); (void)hipMalloc(&cyc, 256 * 8 * sizeof(long long));
    (void)hipMemset(cyc, 0, 256 * 8 * sizeof(long long));
    const int iters = 20000;
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, 100); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((k<MODE, NT>), dim3(256), dim3(NT), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize();
    std::vector<long long> hc(256 * 8); (void)hipMemcpy(hc.data(), cyc, 256 * 8 * sizeof(long long), hipMemcpyDeviceToHost);
    double a = 0, bsum = 0; int na = 0, nb = 0;
    for (int blk = 0; blk < 256; ++blk) for (int w = 0; w < NT / 64; ++w) { if (w < 4) { a += hc[blk * 8 + w]; ++na; } else { bsum += hc[blk * 8 + w]; ++nb; } }
    printf("%-70s waves0-3 %8.1f ticks/iter", name, a / na / iters);
    if (nb) printf("   waves4-7 %8.1f", bsum / nb / iters);
    printf("\n");
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0, 256>("1 wave/SIMD: 64 pk_fma");
    run<0, 512>("2 waves/SIMD: 64 pk_fma each");
    run<1, 512>("2 waves/SIMD: A = 64 pk_fma, B = 6 mfma + 16 pk_fma");
    run<2, 512>("same + s_barrier per iter");
    run<3, 512>("2 waves/SIMD: 64 pk_fma each + s_barrier per iter");
    run<3, 256>("1 wave/SIMD: 64 pk_fma + s_barrier per iter (4 waves)");
    return 0;
}
