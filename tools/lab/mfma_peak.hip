// Bare fp32 MFMA issue-rate probe: what TFLOP/s does this chip sustain on v_mfma_f32_32x32x2_f32 / 16x16x4 with
// operands in registers, random data, 1..8 waves per SIMD, 1..4 independent accumulators?  (tuning aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, const float* in, int iters) {
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float x = in[threadIdx.x], y = in[threadIdx.x + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// same, but every MFMA takes a different random operand pair (data toggling as in a real GEMM)
template <int NACC>
__global__ void k32r(float* out, const float* in, int iters) {
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float x[8], y[8];
    for (int u = 0; u < 8; ++u) { x[u] = in[(threadIdx.x * 7 + u * 13) & 255]; y[u] = in[(threadIdx.x * 5 + u * 29 + 3) & 255]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[(u + a) & 7], acc[a], 0, 0, 0);
        if ((it & 63) == 63) for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] *= 1e-3f;  // keep finite
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k16(float* out, const float* in, int iters) {
    floatx4 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 4; ++i) acc[a][i] = 0.f;
    float x = in[threadIdx.x], y = in[threadIdx.x + 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 4; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int nacc, double flop_per_mfma, int waves_per_simd, float* out, float* in) {
    const int iters = 2000;
    dim3 grid(256 * waves_per_simd), block(256);     // 4 waves per block -> one per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, in, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 5.0 * 256 * waves_per_simd * 4 * (double)iters * 8 * nacc * flop_per_mfma;
    printf("%s nacc=%d waves/SIMD=%d : %.1f TFLOP/s (%.2f ms)\n", name, nacc, waves_per_simd, flops / (ms * 1e-3) / 1e12, ms / 5);
}
int main() {
    float *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 1024);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4}) {
        run("32x32x2", k32<1>, 1, 4096, w, out, in);
        run("32x32x2", k32<2>, 2, 4096, w, out, in);
        run("32x32x2", k32<4>, 4, 4096, w, out, in);
        run("32x32x2 random operands", k32r<1>, 1, 4096, w, out, in);
        run("32x32x2 random operands", k32r<4>, 4, 4096, w, out, in);
        run("16x16x4", k16<1>, 1, 2048, w, out, in);
        run("16x16x4", k16<4>, 4, 2048, w, out, in);
    }
    return 0;
}
