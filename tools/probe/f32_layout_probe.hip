// Operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f32 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
    const int l = threadIdx.x;
    // A[i][kk] = 100*i + kk ... encode so D identifies its (row, col): A row i = [1,0,0,0]*(i+1), B col j: k=0 -> (j+1)*1000
    const int i = l & 15, kk = l >> 4;
    const float a = kk == 0 ? (float)(i + 1) : 0.0f;          // hypothesis: lane holds A[i = l&15][k = l>>4]
    const float b = kk == 0 ? (float)((i + 1) * 1000) : 0.0f;  // hypothesis: lane holds B[k = l>>4][j = l&15]
    v4f32 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    // D[row][col] = (row+1) * (col+1)*1000 under the hypothesis; decode where each (lane, reg) landed
    int ok_f64_like = 1, ok_f32_like = 1;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const float v = h[l * 4 + r];
        const int col = l & 15;
        const int row_a = (l >> 4) + 4 * r;      // f64-style: row = (l>>4) + 4 r
        const int row_b = 4 * (l >> 4) + r;      // f32-style: row = 4 (l>>4) + r
        if (v != (float)((row_a + 1) * (col + 1) * 1000)) ok_f64_like = 0;
        if (v != (float)((row_b + 1) * (col + 1) * 1000)) ok_f32_like = 0;
    }
    printf("row = (l>>4) + 4r: %d   row = 4(l>>4) + r: %d\n", ok_f64_like, ok_f32_like);
    printf("lane 0: %g %g %g %g   lane 16: %g %g %g %g   lane 1: %g %g\n", h[0], h[1], h[2], h[3], h[64], h[65], h[66], h[67], h[4], h[5]);
    return 0;
}
