// Microbenchmark 2: which instruction kinds of the SAME wave hide behind v_mfma_f32_32x32x16_bf16 (one wave per SIMD, 4 waves per CU)?
// Per iteration 8 MFMAs on 4 accumulators; after each, K instructions of one kind (inline asm, independent registers), pinned with sched_barrier.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s8 __attribute__((ext_vector_type(8)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
enum { AND = 0, PERM, CND, MUL, FMA, PKFMA, DSR128, DSW64, GLD, MIX };
template <int KIND, int K>
__global__ __launch_bounds__(256, 1) void k(float* out, const float* in, int iters) {
    __shared__ f4 lds[1024];
    s8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {(short)threadIdx.x, 1, 2, 3, 4, 5, 6, 7};
    f16 c[4];
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) c[j][r] = 0;
    float x[8];
    unsigned y[8];
    f4 q[4];
    float2 p[4];
    for (int j = 0; j < 8; ++j) { x[j] = threadIdx.x * 0.5f + j; y[j] = threadIdx.x + j; }
    for (int j = 0; j < 4; ++j) { q[j] = f4{0, 0, 0, 0}; p[j] = make_float2(1.f, 2.f); }
    lds[threadIdx.x] = f4{1, 2, 3, 4};
    __syncthreads();
    const unsigned la = threadIdx.x * 16;
    const unsigned sel = 0x07060302u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            c[r & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[r & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < K; ++v) {
                const int j = (r + v) & 7;
                const int kind = KIND == MIX ? (v % 4 == 0 ? AND : v % 4 == 1 ? PERM : v % 4 == 2 ? MUL : CND) : KIND;
                if (kind == AND) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(y[j]));
                if (kind == PERM) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(y[j]) : "v"(y[(j + 1) & 7]), "s"(sel));
                if (kind == CND) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x[j]) : "v"(x[(j + 1) & 7]));
                if (kind == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[j]) : "v"(x[(j + 3) & 7]));
                if (kind == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[j]) : "v"(x[(j + 3) & 7]), "v"(x[(j + 5) & 7]));
                if (kind == PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 3]) : "v"(p[(j + 1) & 3]), "v"(p[(j + 2) & 3]));
                if (kind == DSR128) asm volatile("ds_read_b128 %0, %1" : "=v"(q[j & 3]) : "v"(la));
                if (kind == DSW64) asm volatile("ds_write_b64 %0, %1" : : "v"(la), "v"(p[j & 3]));
                if (kind == GLD) asm volatile("global_load_dword %0, %1, off" : "=v"(x[j]) : "v"(in + threadIdx.x));
            }
            if (KIND == DSR128 || KIND == DSW64) asm volatile("s_waitcnt lgkmcnt(0)");
            if (KIND == GLD) asm volatile("s_waitcnt vmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int j = 0; j < 4; ++j) s += c[j][0] + q[j][0] + p[j].x;
    for (int j = 0; j < 8; ++j) s += x[j] + (float)y[j];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
static float* g_out; static float* g_in;
template <typename Kn> void run(const char* name, int K, Kn kern) {
    const int iters = 10000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, g_out, g_in, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, g_out, g_in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s K=%2d : %6.2f ns per MFMA\n", name, K, ms * 1e6 / (iters * 8.0));
}
#define ROW(KIND, name) run(name, 0, k<KIND, 0>); run(name, 2, k<KIND, 2>); run(name, 4, k<KIND, 4>); run(name, 6, k<KIND, 6>); run(name, 8, k<KIND, 8>); run(name, 12, k<KIND, 12>);
int main() {
    (void)hipMalloc(&g_out, 256 * 256 * 4); (void)hipMalloc(&g_in, 4096); (void)hipMemset(g_in, 0, 4096);
    ROW(AND, "v_and_b32") ROW(PERM, "v_perm_b32") ROW(CND, "v_cndmask") ROW(MUL, "v_mul_f32") ROW(FMA, "v_fma_f32") ROW(PKFMA, "v_pk_fma_f32")
    ROW(MIX, "mix") 
    run("ds_read_b128", 1, k<DSR128, 1>); run("ds_read_b128", 2, k<DSR128, 2>); run("ds_read_b128", 4, k<DSR128, 4>);
    run("ds_write_b64", 1, k<DSW64, 1>); run("ds_write_b64", 2, k<DSW64, 2>); run("ds_write_b64", 4, k<DSW64, 4>);
    run("global_load", 1, k<GLD, 1>); run("global_load", 2, k<GLD, 2>); run("global_load", 4, k<GLD, 4>);
    return 0;
}
