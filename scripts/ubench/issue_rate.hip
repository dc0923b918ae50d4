// Microbenchmark: cycles per instruction for one wave per SIMD (256-thread blocks, one per CU).
// Build: hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));
#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    v2f a0 = {1.f, 2.f}, a1 = {3.f, 4.f}, a2 = {5.f, 6.f}, a3 = {7.f, 8.f};
    v2f m = {1.0001f, 0.9999f}, b = {0.5f, 0.25f};
    float f0 = 1.f, f1 = 2.f, f2 = 3.f, f3 = 4.f, f4 = 1.f, f5 = 2.f, f6 = 3.f, f7 = 4.f;
    v16f acc0 = {}, acc1 = {};
    float ma = threadIdx.x * 1e-3f, mb = 1.f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // 64 independent-chain v_pk_fma_f32 (4 chains)
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(b));)
        } else if (MODE == 1) {  // 64 v_fma_f32 (8 chains)
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(mb), "v"(ma));)
        } else if (MODE == 2) {  // pk_fma with op_sel/neg modifiers (as the kernel uses), 2 chains
            REP16(asm volatile("v_pk_fma_f32 %0, %2, %3, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n v_pk_fma_f32 %1, %2, %3, %1 op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                               "v_pk_fma_f32 %0, %2, %3, %0 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n v_pk_fma_f32 %1, %2, %3, %1 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]"
                               : "+v"(a0), "+v"(a1) : "v"(m), "v"(b));)
        } else if (MODE == 3) {  // 64 pk_fma + 2 MFMA 32x32x2 f32 interleaved
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(b));)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
        } else if (MODE == 4) {  // 64 v_fma + 2 MFMA
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(mb), "v"(ma));)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
        } else if (MODE == 5) {  // 64 pk_fma, ONE dependent chain
            REP16(asm volatile("v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0\n v_pk_fma_f32 %0, %1, %2, %0"
                               : "+v"(a0) : "v"(m), "v"(b));)
        } else if (MODE == 6) {  // 64 v_fmac with DPP row_ror (the alternative mat-vec form), 4 chains
            REP16(asm volatile("v_fmac_f32_dpp %0, %4, %5 row_ror:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %1, %4, %5 row_ror:2 row_mask:0xf bank_mask:0xf\n"
                               "v_fmac_f32_dpp %2, %4, %5 row_ror:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %4, %5 row_ror:4 row_mask:0xf bank_mask:0xf"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(mb), "v"(ma));)
        } else if (MODE == 7) {  // 6 MFMA 32x32x2 f32 only (independent accumulators x2)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
        } else if (MODE == 8) {  // 64 pk_fma + 6 MFMA
            REP16(asm volatile("v_pk_fma_f32 %0, %4, %5, %0\n v_pk_fma_f32 %1, %4, %5, %1\n v_pk_fma_f32 %2, %4, %5, %2\n v_pk_fma_f32 %3, %4, %5, %3"
                               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(b));)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
        } else if (MODE == 9) {  // 128 v_fma_f32 + 6 MFMA (same flops as mode 8)
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                               : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(mb), "v"(ma));)
            REP16(asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"
                               : "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(mb), "v"(ma));)
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ma, mb, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(mb, ma, acc1, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float r = a0.x + a1.x + a2.x + a3.x + a0.y + a1.y + a2.y + a3.y + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
    for (int q = 0; q < 16; ++q) r += acc0[q] + acc1[q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int ninstr) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(long long));
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(256);
    hipMemcpy(h.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : h) avg += c; avg /= 256;
    printf("%-44s  %8.1f memtime-ticks/iter  %8.1f ns/iter  (%d instr/iter -> %.2f ticks/instr, %.2f ns/instr)\n", name,
           avg / iters, ms * 1e6 / iters, ninstr, avg / iters / ninstr, ms * 1e6 / iters / ninstr);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("64 pk_fma, 4 chains", 64);
    run<1>("64 v_fma, 4 chains", 64);
    run<2>("64 pk_fma op_sel/neg, 2 chains", 64);
    run<5>("64 pk_fma, 1 chain", 64);
    run<6>("64 v_fmac_dpp row_ror, 4 chains", 64);
    run<7>("6 mfma 32x32x2 f32", 6);
    run<3>("64 pk_fma + 2 mfma", 66);
    run<4>("64 v_fma + 2 mfma", 66);
    run<8>("64 pk_fma + 6 mfma", 70);
    run<9>("128 v_fma + 6 mfma", 134);
    return 0;
}
