// Microbenchmark: LDS read THROUGHPUT per CU for the access patterns of the scan (independent reads, 4 or 8 waves per
// CU, all 256 CUs busy).  Reports nanoseconds and core clocks (2.4 GHz) per wave-level read instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p; }

// MODE 0: ds_read_b128, every lane of a half reads the same 8 x 16 B (the scan's broadcast)
// MODE 1: ds_read_b128, every lane its own 16 B (conflict-free, 1 KB per instruction)
// MODE 2: ds_read_b64 broadcast   MODE 3: ds_read_b32 broadcast
// MODE 4: ds_read_b128 broadcast, all 64 lanes the same address
// MODE 5: ds_read_b128, groups of 8 lanes share an address (8 distinct 16-B blocks per instruction)
template <int MODE, int NW>
__global__ __launch_bounds__(64 * NW, 1) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float buf[NW][2048];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, h = lane >> 5;
    for (int j = lane; j < 2048; j += 64) buf[w][j] = j * 0.001f;
    __syncthreads();
    unsigned rd = lds_addr(&buf[w][0]);
    if (MODE == 0 || MODE == 2 || MODE == 3) rd += h * 128;
    if (MODE == 1) rd += lane * 16;
    if (MODE == 5) rd += (lane >> 3) * 16;
    v4f o[8];
    v2f p[8];
    float f[8];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0 || MODE == 4 || MODE == 5) {
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:16\n ds_read_b128 %2, %8 offset:32\n ds_read_b128 %3, %8 offset:48\n"
                         "ds_read_b128 %4, %8 offset:64\n ds_read_b128 %5, %8 offset:80\n ds_read_b128 %6, %8 offset:96\n ds_read_b128 %7, %8 offset:112\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]) : "v"(rd) : "memory");
        } else if (MODE == 1) {
            asm volatile("ds_read_b128 %0, %8\n ds_read_b128 %1, %8 offset:1024\n ds_read_b128 %2, %8 offset:2048\n ds_read_b128 %3, %8 offset:3072\n"
                         "ds_read_b128 %4, %8 offset:4096\n ds_read_b128 %5, %8 offset:5120\n ds_read_b128 %6, %8 offset:6144\n ds_read_b128 %7, %8 offset:7168\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7]) : "v"(rd) : "memory");
        } else if (MODE == 2) {
            asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:8\n ds_read_b64 %2, %8 offset:16\n ds_read_b64 %3, %8 offset:24\n"
                         "ds_read_b64 %4, %8 offset:32\n ds_read_b64 %5, %8 offset:40\n ds_read_b64 %6, %8 offset:48\n ds_read_b64 %7, %8 offset:56\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(p[0]), "=&v"(p[1]), "=&v"(p[2]), "=&v"(p[3]), "=&v"(p[4]), "=&v"(p[5]), "=&v"(p[6]), "=&v"(p[7]) : "v"(rd) : "memory");
            o[0].x = p[0].x; o[7].x = p[7].x;
        } else {
            asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n"
                         "ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n s_waitcnt lgkmcnt(0)"
                         : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]), "=&v"(f[4]), "=&v"(f[5]), "=&v"(f[6]), "=&v"(f[7]) : "v"(rd) : "memory");
            o[0].x = f[0]; o[7].x = f[7];
        }
        acc += o[0].x + o[7].x;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE, int NW> void run(const char* name) {
    float* out;
    (void)hipMalloc(&out, 256 * 64 * NW * sizeof(float));
    const int iters = 40000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NW>), dim3(256), dim3(64 * NW), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NW>), dim3(256), dim3(64 * NW), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 8 * NW);        // per wave-level read instruction per CU
    printf("%-58s %d waves/CU: %6.2f ns = %5.1f clk @2.4GHz per read instr per CU\n", name, NW, ns, ns * 2.4);
    (void)hipFree(out);
}
int main() {
    run<0, 4>("b128 broadcast per half (the scan's pattern)");
    run<0, 8>("b128 broadcast per half (the scan's pattern)");
    run<4, 4>("b128, all 64 lanes one address");
    run<5, 4>("b128, 8 lanes per address");
    run<1, 4>("b128, one 16-B block per lane (1 KB)");
    run<1, 8>("b128, one 16-B block per lane (1 KB)");
    run<2, 4>("b64 broadcast per half");
    run<2, 8>("b64 broadcast per half");
    run<3, 4>("b32 broadcast per half");
    run<3, 8>("b32 broadcast per half");
    return 0;
}
