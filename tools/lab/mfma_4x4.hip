// Layout probe for v_mfma_f32_4x4x1_16B_f32 (16 independent 4x4x1 outer products per instruction): which lane feeds which
// (block, row / column) and where the results land.  Used to plan the "96 + 4 columns" decomposition of the 100-wide
// dimension.  Prints the inferred mapping.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float floatx4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, const float* a, const float* b) {
    floatx4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[threadIdx.x], b[threadIdx.x], acc, 0, 0, 0);
    for (int i = 0; i < 4; ++i) out[threadIdx.x * 4 + i] = acc[i];
}
int main() {
    float ha[64], hb[64], ho[256];
    float *a, *b, *o; hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&o, 1024);
    // a[lane] = 1 + lane (distinct), b[lane] = prime-ish distinct -> every product identifies its (a lane, b lane)
    for (int l = 0; l < 64; ++l) { ha[l] = 1.f + l; hb[l] = 101.f + 2 * l; }
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, a, b);
    hipMemcpy(ho, o, 1024, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l)
        for (int r = 0; r < 4; ++r) {
            const float v = ho[l * 4 + r];
            // hypothesis: block = l / 4; D[lane = 4 blk + j][reg i] = A[lane 4 blk + i] * B[lane 4 blk + j]
            if (v != ha[4 * (l / 4) + r] * hb[l]) ok = 0;
        }
    printf("hypothesis D[4b+j][i] = A[4b+i] * B[4b+j]: %s\n", ok ? "CONFIRMED" : "not confirmed");
    return 0;
}
