// Operand layout of v_mfma_f64_16x16x4_f64 on gfx950, probed: D = A (16 x 4) * B (4 x 16).
// Prints, for every lane and result register, which (row, column) of D it holds, and checks the
// assumed A / B placement: A[i = lane % 16][k = lane / 16], B[k = lane / 16][j = lane % 16],
// D[i = 4 (lane / 16) + r][j = lane % 16].
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f64_layout profiles/micro/mfma_f64_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double *A, const double *B, double *D) {
    const int l = threadIdx.x;
    const double a = A[(l % 16) * 4 + l / 16];      // A[i][k], row-major 16 x 4
    const double b = B[(l / 16) * 16 + l % 16];     // B[k][j], row-major 4 x 16
    v4d c = {0.0, 0.0, 0.0, 0.0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = c[r];              // raw: [lane][register]
}
int main() {
    double hA[64], hB[64], hD[256], ref[256];
    for (int i = 0; i < 64; ++i) { hA[i] = 1.0 + 0.37 * i; hB[i] = -2.0 + 0.11 * i * i; }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 4; ++k) s += hA[i * 4 + k] * hB[k * 16 + j];
        ref[i * 16 + j] = s;
    }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dD, sizeof hD);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost);
    // two candidate placements of D: rows 4 (lane / 16) + r, or rows 4 r + lane / 16
    double e1 = 0, e2 = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const double v = hD[l * 4 + r];
        const double a = ref[(4 * (l / 16) + r) * 16 + l % 16], b = ref[(4 * r + l / 16) * 16 + l % 16];
        e1 = fmax(e1, fabs(v - a) / fabs(a)); e2 = fmax(e2, fabs(v - b) / fabs(b));
    }
    printf("D[4 (lane/16) + r][lane%%16]: max relative error %.3g\n", e1);
    printf("D[4 r + lane/16][lane%%16]: max relative error %.3g\n", e2);
    if (e1 > 1e-14 && e2 > 1e-14)
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            int hit = -1;
            for (int q = 0; q < 256; ++q) if (fabs(hD[l * 4 + r] - ref[q]) <= 1e-13 * fabs(ref[q])) hit = q;
            printf("lane %d reg %d -> D[%d][%d]\n", l, r, hit / 16, hit % 16);
        }
    return 0;
}
