// Microbenchmark: does the VGPR bank (register number mod 4) of the operands of v_pk_fma_f32 matter for a lone wave?
// The mat-vec chains of the wave kernels are  v_pk_fma_f32 acc, M, b, acc  with op_sel broadcasts of b; hipcc assigns M and b freely.
// Literal registers: acc = v[20:21] (bank pair 0) or v[22:23] (pair 2); M, b in pair 0 (v[4k]) or pair 2 (v[4k+2]).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, int iters) {
    float s = 0.f;
    // initialise v0..v31 through asm so the compiler leaves them alone (clobbers declared)
    asm volatile("v_mov_b32 v4, 1.0\n v_mov_b32 v5, 0.5\n v_mov_b32 v6, 0.25\n v_mov_b32 v7, 2.0\n"
                 "v_mov_b32 v8, 1.0\n v_mov_b32 v9, 0.5\n v_mov_b32 v10, 0.25\n v_mov_b32 v11, 2.0\n"
                 "v_mov_b32 v12, 1.0\n v_mov_b32 v13, 0.5\n v_mov_b32 v14, 0.25\n v_mov_b32 v15, 2.0\n"
                 "v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
                 ::: "v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v20","v21","v22","v23");
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)       // acc pair 0, M pair 0, b pair 0: all three in the same bank pair
            asm volatile(REP8("v_pk_fma_f32 v[20:21], v[4:5], v[8:9], v[20:21] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 v[20:21], v[4:5], v[8:9], v[20:21] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n")
                         ::: "v20","v21");
        else if (MODE == 1)  // acc pair 0, M pair 2, b pair 0
            asm volatile(REP8("v_pk_fma_f32 v[20:21], v[6:7], v[8:9], v[20:21] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 v[20:21], v[6:7], v[8:9], v[20:21] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n")
                         ::: "v20","v21");
        else if (MODE == 2)  // acc pair 2, M pair 0, b pair 0
            asm volatile(REP8("v_pk_fma_f32 v[22:23], v[4:5], v[8:9], v[22:23] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 v[22:23], v[4:5], v[8:9], v[22:23] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n")
                         ::: "v22","v23");
        else if (MODE == 3)  // acc pair 2, M pair 0, b pair 2
            asm volatile(REP8("v_pk_fma_f32 v[22:23], v[4:5], v[10:11], v[22:23] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 v[22:23], v[4:5], v[10:11], v[22:23] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n")
                         ::: "v22","v23");
        else if (MODE == 4)  // scalar v_fma_f32 x 2 per pk (same flops): all different banks
            asm volatile(REP8("v_fma_f32 v20, v4, v9, v20\n v_fma_f32 v21, v5, v10, v21\n v_fma_f32 v22, v6, v11, v22\n v_fma_f32 v23, v7, v8, v23\n")
                         ::: "v20","v21","v22","v23");
        else                 // independent accumulators (no dependent chain), mixed banks
            asm volatile(REP8("v_pk_fma_f32 v[20:21], v[6:7], v[8:9], v[20:21] op_sel:[0,0,0] op_sel_hi:[1,0,1]\n"
                              "v_pk_fma_f32 v[22:23], v[4:5], v[10:11], v[22:23] op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n")
                         ::: "v20","v21","v22","v23");
    }
    asm volatile("v_add_f32 %0, v20, v22" : "=v"(s) :: );
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <typename K> void run(const char* name, K kern, int per) {
    float* out; (void)hipMalloc(&out, 256 * 256 * 4);
    const int iters = 100000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, 100); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-72s %6.2f ns per instruction\n", name, ms * 1e6 / ((double)iters * per));
    (void)hipFree(out);
}
int main() {
    run("v_pk_fma_f32 acc(0) M(0) b(0): one bank pair, dependent chain", k<0>, 16);
    run("v_pk_fma_f32 acc(0) M(2) b(0)", k<1>, 16);
    run("v_pk_fma_f32 acc(2) M(0) b(0)", k<2>, 16);
    run("v_pk_fma_f32 acc(2) M(0) b(2)", k<3>, 16);
    run("v_fma_f32, four banks (per scalar instruction)", k<4>, 32);
    run("v_pk_fma_f32 two independent accumulators, mixed banks", k<5>, 16);
    return 0;
}
