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
__global__ void rate(float* out, const float* in, int iters) {
    floatx4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    float x[4], y[4];
    for (int u = 0; u < 4; ++u) { x[u] = in[(threadIdx.x + u) & 63]; y[u] = in[(threadIdx.x * 3 + u) & 63]; }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_4x4x1f32(x[u & 3], y[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
    float s = 0.f;
    for (int a = 0; a < 4; ++a) for (int i = 0; i < 4; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static void rate_probe(float* o, const float* a) {
    for (int w : {1, 2, 4}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int iters = 4000;
        hipLaunchKernelGGL(rate, dim3(256 * w), dim3(256), 0, 0, o, a, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(rate, dim3(256 * w), dim3(256), 0, 0, o, a, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 3.0 * 256 * w * 4 * (double)iters * 16 * 512;
        printf("4x4x1 MFMA, 4 accumulators, %d waves/SIMD: %.1f TFLOP/s (157.3 = the fp32 MFMA peak)\n", w, flops / (ms * 1e-3) / 1e12);
    }
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
    { float* big; hipMalloc(&big, 256 * 4 * 256 * 4 * sizeof(float)); rate_probe(big, a); }
    printf("hypothesis D[4b+j][i] = A[4b+i] * B[4b+j]: %s\n", ok ? "CONFIRMED" : "not confirmed");
    return 0;
}
