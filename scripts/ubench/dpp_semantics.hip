// Semantics check for the DPP controls the D <= 16 kernels' register mat-vec relies on (cmps_wave16.hip): direction of row_ror,
// row_mask on an in-place move, row_newbcast with a neg modifier on the DPP source.  Prints the lanes' values.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float x = (float)lane, y = (float)lane;
    // row q rotated "right" by 4 q
    asm volatile("s_nop 1\n"
                 "v_mov_b32_dpp %0, %0 row_ror:4 row_mask:0x2 bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0x4 bank_mask:0xf\n"
                 "s_nop 1\n"
                 "v_mov_b32_dpp %0, %0 row_ror:12 row_mask:0x8 bank_mask:0xf\n"
                 "s_nop 1\n" : "+v"(x));
    float acc = 100.f, m = 2.f;
    asm volatile("s_nop 1\n"
                 "v_fmac_f32_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                 "s_nop 1\n" : "+v"(acc) : "v"(y), "v"(m));
    out[lane] = x;
    out[64 + lane] = acc;
}
int main() {
    float* d; (void)hipMalloc(&d, 128 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    float h[128]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int q = 0; q < 4; ++q) { printf("row %d after row_ror:%d  :", q, 4 * q); for (int i = 0; i < 16; ++i) printf(" %2.0f", h[16 * q + i]); printf("\n"); }
    for (int q = 0; q < 4; ++q) { printf("row %d  100 - 2 * lane[3 of row]:", q); for (int i = 0; i < 16; i += 5) printf(" %4.0f", h[64 + 16 * q + i]); printf("\n"); }
    return 0;
}
