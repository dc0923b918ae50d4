// Microbenchmark for VERDICT r3 item 5 (C2, D <= 16): a K-quarter mat-vec as v_fmac_f32 with DPP row_newbcast (lane n of the own
// 16-lane row, no LDS) against the LDS-broadcast form (ds_write_b32 + ds_read_b128 + 8 v_pk_fma_f32), from a lone wave per SIMD
// with a dependent recurrence (the output of one "step" is the input of the next, as in the scan).
//   A: 16 v_fmac_f32_dpp row_newbcast (4 complex entries: re*re, -im*im, re*im, im*re) + redistribution of the result into the
//      rows' broadcast lanes: 2 x 4 v_mov_b32_dpp row_ror:4q with row_mask (row q rotates by 4 q) + the two cross-row combines
//   B: ds_write_b32, s_waitcnt, 2 ds_read_b128, s_waitcnt, 8 v_pk_fma_f32, the same two combines
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float m[16];
    for (int i = 0; i < 16; ++i) m[i] = 0.01f * ((threadIdx.x * 7 + i * 13) % 17) - 0.08f;
    float xr = 0.1f + 0.001f * lane, xi = 0.05f;
    const unsigned aw = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[w][lane];
    const unsigned ar = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void*)&lds[w][(lane >> 4) * 8];
    for (int it = 0; it < iters; ++it) {
        float ar_ = xr, ai_ = xi;
        if (MODE == 0) {
            // redistribution: row q takes its inputs from lanes 4 q .. 4 q + 3 -> rotate row q by 4 q (masked DPP moves)
            asm volatile("s_nop 1\n"
                         "v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0x2 bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_ror:4 row_mask:0x2 bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0x4 bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_ror:8 row_mask:0x4 bank_mask:0xf\n"
                         "v_mov_b32_dpp %0, %0 row_ror:12 row_mask:0x8 bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_ror:12 row_mask:0x8 bank_mask:0xf\n"
                         "s_nop 1\n"
                         "v_mul_f32_dpp %2, %0, %4 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %0, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %1, %6 row_newbcast:0 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %1, %7 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %0, %9 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %1, %10 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %1, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %0, %12 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %0, %13 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %1, %14 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %1, %15 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %0, %16 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %0, %17 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         "v_fmac_f32_dpp %2, %1, %18 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f32_dpp %3, %1, %19 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(xr), "+v"(xi), "=&v"(ar_), "=&v"(ai_)
                         : "v"(m[0]), "v"(m[1]), "v"(m[2]), "v"(m[3]), "v"(m[4]), "v"(m[5]), "v"(m[6]), "v"(m[7]), "v"(m[8]), "v"(m[9]),
                           "v"(m[10]), "v"(m[11]), "v"(m[12]), "v"(m[13]), "v"(m[14]), "v"(m[15]));
        } else {
            float4 q0, q1;
            asm volatile("ds_write_b32 %2, %3\n s_waitcnt lgkmcnt(0)\n ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:16\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(q0), "=&v"(q1) : "v"(aw), "v"(xr), "v"(ar) : "memory");
            ar_ = m[0] * q0.x - m[1] * q0.y + m[2] * q0.z - m[3] * q0.w + m[4] * q1.x - m[5] * q1.y + m[6] * q1.z - m[7] * q1.w;
            ai_ = m[8] * q0.y + m[9] * q0.x + m[10] * q0.w + m[11] * q0.z + m[12] * q1.y + m[13] * q1.x + m[14] * q1.w + m[15] * q1.z;
        }
        // the two cross-row combines (sum over the four K quarters), both components
        {
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(ar_), __float_as_uint(ai_), false, false);
            float a = __uint_as_float(r[0]) + __uint_as_float(r[1]);
            auto r2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(a), false, false);
            xr = 0.5f * (__uint_as_float(r2[0]) + __uint_as_float(r2[1]));
            xi = xr * 0.5f + 0.01f;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = xr + xi;
}
template <typename K> void run(const char* name, K kern) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 200000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 1000); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-84s %7.1f ns per step\n", name, ms * 1e6 / iters);
    (void)hipFree(out);
}
int main() {
    run("A: 6 masked row_ror moves + 16 v_fmac_f32_dpp row_newbcast + combines (no LDS)", k<0>);
    run("B: ds_write_b32 -> 2 ds_read_b128 -> 16 scalar FMAs (compiler) + combines (LDS round trip)", k<1>);
    return 0;
}
