// Microbenchmark for the round-4 pair kernels: v_mfma_f32_16x16x32_bf16 as a mat-vec engine with FOUR useful A rows.
//   (1) layout check with exact integer data: A rows {0, 4, 8, 12} carry the four vector forms, B = 16 matrix rows (K along
//       the lane groups); D register 0 of lane l must then be  form (l >> 4) . matrix row (l & 15)  -- one useful value in every
//       lane, no cross-lane K reduction, no compaction.
//   (2) issue rate of the instruction from a lone wave per SIMD and from two, bare / with LDS operand reads / with VALU
//       behind each MFMA (round 2's scripts/ubench/mfma4_rate.hip timed a loop that hipcc had filled with 25 accvgpr moves
//       per 8 MFMAs: its "32 clk" for this instruction was the moves, not the matrix pipe).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

static unsigned short bf16_of(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }   // exact for small ints

__global__ void k_layout(const unsigned short* A, const unsigned short* B, float* D) {
    // A: [16][32] row-major, B: [32][16] row-major (K x N); D out: [64 lanes][4 regs]
    const int l = threadIdx.x, i = l & 15, kg = l >> 4;
    s8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (short)A[i * 32 + 8 * kg + j]; b[j] = (short)B[(8 * kg + j) * 16 + i]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];
}

// NW waves per workgroup (256: one per SIMD, 512: two); per iteration 32 MFMAs on four accumulators in rotation.
// MODE 0: bare.  MODE 1: + 8 ds_read_b128 per iteration feeding the A operands (counted waits).  MODE 2: MODE 1 + VPM VALU behind each MFMA.
template <int NT, int MODE, int VPM>
__global__ __launch_bounds__(NT, 1) void k_rate(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char vec[8 * 1024];
    for (int i = threadIdx.x; i < 2048; i += NT) reinterpret_cast<unsigned*>(vec)[i] = 0x3f803f80u + (i & 7);
    __syncthreads();
    const unsigned addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)vec + (threadIdx.x & 60) * 16;
    u4 bfrag = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x};
    u4 a0 = bfrag, a1 = bfrag, a2 = bfrag, a3 = bfrag, a4 = bfrag, a5 = bfrag, a6 = bfrag, a7 = bfrag;
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = threadIdx.x, x1 = 1.0001f, x2 = 0.5f, x3 = 2.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 1) {
            asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                         "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                         "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(addr) : "memory");
        }
#define VAL() if (MODE == 2) { _Pragma("unroll") for (int v = 0; v < VPM; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(x1), "v"(x2)); \
                               asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x3) : "v"(x1)); }
#define STEP(A, W)                                                                                              \
        asm volatile("s_waitcnt lgkmcnt(" #W ")\n\t"                                                           \
                     "v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(A), "v"(bfrag)); VAL() \
        asm volatile("v_mfma_f32_16x16x32_bf16 %1, %4, %5, %1\n\t" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(A), "v"(bfrag)); VAL() \
        asm volatile("v_mfma_f32_16x16x32_bf16 %2, %4, %5, %2\n\t" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(A), "v"(bfrag)); VAL() \
        asm volatile("v_mfma_f32_16x16x32_bf16 %3, %4, %5, %3\n\t" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(A), "v"(bfrag)); VAL()
        if (MODE >= 1) { STEP(a0, 7) STEP(a1, 6) STEP(a2, 5) STEP(a3, 4) STEP(a4, 3) STEP(a5, 2) STEP(a6, 1) STEP(a7, 0) }
        else { STEP(a0, 0) STEP(a1, 0) STEP(a2, 0) STEP(a3, 0) STEP(a4, 0) STEP(a5, 0) STEP(a6, 0) STEP(a7, 0) }
#undef STEP
#undef VAL
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    out[blockIdx.x * NT + threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + x0 + x3;
}


// B operands (eight distinct fragments) resident in AGPRs ("a"), A from VGPRs, accumulators in VGPRs -- the pair kernels' operand mix.
// VIN VALU instructions are written INSIDE the statement behind each MFMA (no hipcc s_nop pad between them).
template <int NT, int VIN>
__global__ __launch_bounds__(NT, 1) void k_rate_a(float* out, int iters) {
    u4 bf[8];
    for (int i = 0; i < 8; ++i) { bf[i] = u4{0x3f803f80u, 0x3f803f80u + i, 0x3f803f80u, threadIdx.x}; asm volatile("" : "+a"(bf[i])); }
    u4 a0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x};
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = threadIdx.x, x1 = 1.0001f, x2 = 0.5f, x3 = 2.f;
    for (int it = 0; it < iters; ++it) {
#define V1 "v_fma_f32 %9, %9, %10, %11\n\t"
#define V2 "v_fma_f32 %9, %9, %10, %11\n\tv_mul_f32 %12, %12, %10\n\t"
#define V3 "v_fma_f32 %9, %9, %10, %11\n\tv_mul_f32 %12, %12, %10\n\tv_fma_f32 %9, %9, %10, %11\n\t"
#define KST(F0, F1, F2, F3, VV)                                                                                  \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\t" VV "v_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t" VV \
                     "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\t" VV "v_mfma_f32_16x16x32_bf16 %3, %4, %8, %3\n\t" VV \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "a"(F0), "a"(F1), "a"(F2), "a"(F3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        if (VIN == 0) { for (int r = 0; r < 4; ++r) { KST(bf[0], bf[1], bf[2], bf[3], "") KST(bf[4], bf[5], bf[6], bf[7], "") } }
        else if (VIN == 1) { for (int r = 0; r < 4; ++r) { KST(bf[0], bf[1], bf[2], bf[3], V1) KST(bf[4], bf[5], bf[6], bf[7], V1) } }
        else if (VIN == 2) { for (int r = 0; r < 4; ++r) { KST(bf[0], bf[1], bf[2], bf[3], V2) KST(bf[4], bf[5], bf[6], bf[7], V2) } }
        else { for (int r = 0; r < 4; ++r) { KST(bf[0], bf[1], bf[2], bf[3], V3) KST(bf[4], bf[5], bf[6], bf[7], V3) } }
#undef KST
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    out[blockIdx.x * NT + threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + x0 + x3;
}
// the same 32 MFMAs + 64 VALU per iteration with the VALU in blocks of eight BEHIND each K-step of four MFMAs (what compiler-placed
// "pieces" behind an asm K-step amount to)
template <int NT>
__global__ __launch_bounds__(NT, 1) void k_rate_blocks(float* out, int iters) {
    u4 bf[8];
    for (int i = 0; i < 8; ++i) { bf[i] = u4{0x3f803f80u, 0x3f803f80u + i, 0x3f803f80u, threadIdx.x}; asm volatile("" : "+a"(bf[i])); }
    u4 a0 = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, threadIdx.x};
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float x0 = threadIdx.x, x1 = 1.0001f, x2 = 0.5f, x3 = 2.f;
    for (int it = 0; it < iters; ++it) {
#define KSB(F0, F1, F2, F3)                                                                                      \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %5, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %4, %6, %1\n\t"      \
                     "v_mfma_f32_16x16x32_bf16 %2, %4, %7, %2\n\tv_mfma_f32_16x16x32_bf16 %3, %4, %8, %3\n\t"      \
                     "v_fma_f32 %9, %9, %10, %11\n\tv_mul_f32 %12, %12, %10\n\tv_fma_f32 %9, %9, %10, %11\n\tv_mul_f32 %12, %12, %10\n\t" \
                     "v_fma_f32 %9, %9, %10, %11\n\tv_mul_f32 %12, %12, %10\n\tv_fma_f32 %9, %9, %10, %11\n\tv_mul_f32 %12, %12, %10\n\t" \
                     : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a0), "a"(F0), "a"(F1), "a"(F2), "a"(F3), "v"(x0), "v"(x1), "v"(x2), "v"(x3));
        for (int r = 0; r < 4; ++r) { KSB(bf[0], bf[1], bf[2], bf[3]) KSB(bf[4], bf[5], bf[6], bf[7]) }
#undef KSB
    }
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    out[blockIdx.x * NT + threadIdx.x] = c0[0] + c1[0] + c2[0] + c3[0] + x0 + x3;
}

template <typename K> void run(const char* name, K kern, int nt) {
    float* out; (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, 200); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(nt), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = ms * 1e6 / (iters * 32.0 * (nt / 256));
    printf("%-66s %6.2f ns per MFMA per SIMD  (%6.1f ns per 32)\n", name, per_simd, per_simd * 32);
    (void)hipFree(out);
}

int main() {
    // ---- (1) layout ----
    std::vector<unsigned short> A(16 * 32, 0), B(32 * 16);
    std::vector<float> Af(16 * 32, 0.f), Bf(32 * 16);
    srand(5);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 32; ++k) { const float v = (float)(rand() % 7 - 3); Af[i * 32 + k] = v; A[i * 32 + k] = bf16_of(v); }
    for (int k = 0; k < 32; ++k) for (int n = 0; n < 16; ++n) { const float v = (float)(rand() % 9 - 4); Bf[k * 16 + n] = v; B[k * 16 + n] = bf16_of(v); }
    unsigned short *dA, *dB; float* dD;
    (void)hipMalloc(&dA, A.size() * 2); (void)hipMalloc(&dB, B.size() * 2); (void)hipMalloc(&dD, 256 * 4);
    (void)hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    std::vector<float> D(256);
    (void)hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r, col = l & 15;
        float ref = 0; for (int k = 0; k < 32; ++k) ref += Af[row * 32 + k] * Bf[k * 16 + col];
        if (ref != D[l * 4 + r]) ++bad;
    }
    printf("layout 16x16x32 bf16: A[l&15][8(l>>4)+j], B[8(l>>4)+j][l&15], D reg r of lane l = row 4(l>>4)+r, col l&15: %s (%d mismatches)\n",
           bad ? "WRONG" : "confirmed", bad);
    printf("  => forms in A rows {0,4,8,12}: D reg 0 of lane l = form (l>>4) . matrix row (l&15)\n");
    // ---- (2) rates ----
    run("16x16x32 bf16 bare, one wave per SIMD", k_rate<256, 0, 0>, 256);
    run("16x16x32 bf16 bare, two waves per SIMD", k_rate<512, 0, 0>, 512);
    run("16x16x32 bf16 + 8 ds_read_b128 per 32, one wave per SIMD", k_rate<256, 1, 0>, 256);
    run("16x16x32 bf16 + 8 ds_read_b128 per 32, two waves per SIMD", k_rate<512, 1, 0>, 512);
    run("16x16x32 bf16 + reads + 2 VALU per MFMA, one wave per SIMD", k_rate<256, 2, 1>, 256);
    run("16x16x32 bf16 + reads + 3 VALU per MFMA, one wave per SIMD", k_rate<256, 2, 2>, 256);
    run("16x16x32 bf16 + reads + 4 VALU per MFMA, one wave per SIMD", k_rate<256, 2, 3>, 256);
    run("16x16x32 bf16 + reads + 2 VALU per MFMA, two waves per SIMD", k_rate<512, 2, 1>, 512);
    run("16x16x32, B in AGPRs, bare, one wave per SIMD", k_rate_a<256, 0>, 256);
    run("16x16x32, B in AGPRs, 1 VALU inside behind each MFMA, one wave", k_rate_a<256, 1>, 256);
    run("16x16x32, B in AGPRs, 2 VALU inside behind each MFMA, one wave", k_rate_a<256, 2>, 256);
    run("16x16x32, B in AGPRs, 3 VALU inside behind each MFMA, one wave", k_rate_a<256, 3>, 256);
    run("16x16x32, B in AGPRs, 8 VALU behind each block of 4 MFMAs, one wave", k_rate_blocks<256>, 256);
    run("16x16x32, B in AGPRs, 2 VALU inside behind each MFMA, two waves", k_rate_a<512, 2>, 512);
    run("16x16x32, B in AGPRs, 8 VALU behind each block of 4 MFMAs, two waves", k_rate_blocks<512>, 512);
    return 0;
}
