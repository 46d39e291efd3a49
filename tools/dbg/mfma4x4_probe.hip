// Probe: operand layout and numerics of v_mfma_f32_4x4x1_16b_f32 on gfx950.
// Claim to check: lane l = 4 b + r supplies A_b[i = r] and B_b[j = r]; register i of lane l
// receives D_b[i][j = r]; four chained instructions equal the fmaf chain
// fma(a3,b3, fma(a2,b2, fma(a1,b1, a0*b0))) bit for bit.
// build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off mfma4x4_probe.hip -o mfma4x4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f4v __attribute__((ext_vector_type(4)));

__global__ void probe(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f4v c = {0, 0, 0, 0};
    for (int k = 0; k < 4; k++) c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[k * 64 + l], b[k * 64 + l], c, 0, 0, 0);
    for (int i = 0; i < 4; i++) d[l * 4 + i] = c[i];
}

int main() {
    float ha[256], hb[256], hd[256];
    srand(7);
    for (int i = 0; i < 256; i++) {
        ha[i] = (float)rand() / RAND_MAX * 2 - 1;
        hb[i] = ((float)rand() / RAND_MAX * 2 - 1) * (i % 3 == 0 ? 1e4f : 1e-3f);
    }
    float *da, *db, *dd;
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dd, sizeof hd);
    hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
    hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(da, db, dd);
    hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
    int bad = 0, bad_rev = 0;
    for (int l = 0; l < 64; l++)
        for (int i = 0; i < 4; i++) {
            const int blk = l / 4;
            float t = 0.0f, r = 0.0f;
            for (int k = 0; k < 4; k++) t = fmaf(ha[k * 64 + blk * 4 + i], hb[k * 64 + l], t);
            for (int k = 3; k >= 0; k--) r = fmaf(ha[k * 64 + blk * 4 + i], hb[k * 64 + l], r);
            if (memcmp(&t, &hd[l * 4 + i], 4)) bad++;
            if (memcmp(&r, &hd[l * 4 + i], 4)) bad_rev++;
        }
    printf("mismatches vs k-ascending fmaf chain: %d of 256 (k-descending: %d)\n", bad, bad_rev);
    return bad != 0;
}
