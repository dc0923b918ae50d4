// Microbenchmark: latency of the LDS broadcast round trip and cost of the cross-lane idioms used by the scan,
// one wave per SIMD (256-thread blocks, one per CU, all 256 CUs busy).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float2 buf[4][64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5, i = lane & 31;
    const unsigned wr = lds_addr(&buf[w][0]) + i * 8 + h * 4, rd = lds_addr(&buf[w][0]) + h * 128;
    float x = lane * 0.001f + 1.0f;
    v4f o[8];
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {  // broadcast round trip, fully dependent: write, 8 reads, wait, combine into next value
            asm volatile("ds_write_b32 %8, %9\n ds_read_b128 %0, %10\n ds_read_b128 %1, %10 offset:16\n ds_read_b128 %2, %10 offset:32\n ds_read_b128 %3, %10 offset:48\n"
                         "ds_read_b128 %4, %10 offset:64\n ds_read_b128 %5, %10 offset:80\n ds_read_b128 %6, %10 offset:96\n ds_read_b128 %7, %10 offset:112\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]) : "v"(wr), "v"(x), "v"(rd) : "memory");
            x = o[7].w * 0.5f + 0.25f;
        } else if (MODE == 1) {  // write + 1 read round trip
            asm volatile("ds_write_b32 %1, %2\n ds_read_b128 %0, %3\n s_waitcnt lgkmcnt(0)" : "=&v"(o[0]) : "v"(wr), "v"(x), "v"(rd) : "memory");
            x = o[0].w * 0.5f + 0.25f;
        } else if (MODE == 2) {  // sum64 DPP chain as in the kernel
            float y = x;
            asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n s_nop 1\n"
                         "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n s_nop 1\n"
                         "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n s_nop 1\n" : "+v"(y));
            float s = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));
            x = s * 1e-3f + 0.5f;
        } else if (MODE == 3) {  // permlane32 swap + add (swapadd), dependent
            float a = x, b2 = x * 0.5f;
            auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b2), false, false);
            x = (__uint_as_float(r[0]) + __uint_as_float(r[1])) * 0.5f;
        } else if (MODE == 4) {  // v_rsq + NR
            float r = __builtin_amdgcn_rsqf(x);
            x = r * (1.5f - 0.5f * x * r * r) + 0.5f;
        } else if (MODE == 5) {  // broadcast with split wait: first 4 reads then FMA-ish use, then rest
            asm volatile("ds_write_b32 %8, %9\n ds_read_b128 %0, %10\n ds_read_b128 %1, %10 offset:16\n ds_read_b128 %2, %10 offset:32\n ds_read_b128 %3, %10 offset:48\n"
                         "ds_read_b128 %4, %10 offset:64\n ds_read_b128 %5, %10 offset:80\n ds_read_b128 %6, %10 offset:96\n ds_read_b128 %7, %10 offset:112\n s_waitcnt lgkmcnt(4)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]) : "v"(wr), "v"(x), "v"(rd) : "memory");
            x = o[3].w * 0.5f + 0.25f;
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7]));
            x += o[7].x * 1e-9f;
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x + o[0].x;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * sizeof(float)); (void)hipMalloc(&cyc, 256 * sizeof(long long));
    const int iters = 20000;
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, 100); (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize();
    std::vector<long long> hc(256); (void)hipMemcpy(hc.data(), cyc, 256 * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0; for (auto c : hc) avg += c; avg /= 256;
    printf("%-60s %8.1f ticks/iter\n", name, avg / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    run<0>("LDS broadcast round trip (write + 8 b128 + wait), 4 waves/CU");
    run<5>("same, wait for the first 4 reads only, then the rest");
    run<1>("LDS write + 1 b128 read round trip");
    run<2>("sum64 (6 DPP adds with nops + readlane + 1 fma)");
    run<3>("swapadd (permlane32_swap + add + mul)");
    run<4>("rsq + Newton (5 ops)");
    return 0;
}
