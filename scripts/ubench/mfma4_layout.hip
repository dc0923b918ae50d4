// Checks the operand layout of v_mfma_f32_4x4x4_16b_bf16 assumed by cmps_pair.hip:
//   16 blocks of 4 lanes; A: lane (b, r) holds A_b[r][0..3]; B: lane (b, c) holds B_b[0..3][c];
//   D: lane (b, c) register r = D_b[r][c] = sum_k A_b[r][k] B_b[k][c].
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
typedef short s4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
__global__ void k(float* o, const unsigned short* a, const unsigned short* b) {
    s4 A, B;
    for (int j = 0; j < 4; ++j) { A[j] = (short)a[threadIdx.x * 4 + j]; B[j] = (short)b[threadIdx.x * 4 + j]; }
    f4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(A, B, c, 0, 0, 0);
    for (int j = 0; j < 4; ++j) o[threadIdx.x * 4 + j] = c[j];
}
int main() {
    float Af[16][4][4], Bf[16][4][4];   // [block][row][k], [block][k][col]
    unsigned short ha[256], hb[256];
    for (int b = 0; b < 16; ++b) for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        Af[b][i][j] = (float)((b * 7 + i * 3 + j * 5) % 11 - 5);
        Bf[b][i][j] = (float)((b * 5 + i * 2 + j * 9) % 13 - 6);
    }
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) {
        ha[l * 4 + j] = f2bf(Af[l >> 2][l & 3][j]);      // lane (b, r): A_b[r][j]
        hb[l * 4 + j] = f2bf(Bf[l >> 2][j][l & 3]);      // lane (b, c): B_b[j][c]
    }
    unsigned short *da, *db; float* dout; float ho[256];
    (void)hipMalloc(&da, 512); (void)hipMalloc(&db, 512); (void)hipMalloc(&dout, 1024);
    (void)hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); (void)hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dout, da, db);
    (void)hipMemcpy(ho, dout, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        float ref = 0; for (int kk = 0; kk < 4; ++kk) ref += Af[l >> 2][r][kk] * Bf[l >> 2][kk][l & 3];
        if (fabsf(ref - ho[l * 4 + r]) > 1e-3f) { if (bad < 8) printf("lane %d reg %d: got %g want %g\n", l, r, ho[l * 4 + r], ref); ++bad; }
    }
    printf("mfma 4x4x4_16b_bf16 layout check: %s (%d mismatches)\n", bad ? "FAILED" : "ok", bad);
    return bad != 0;
}
