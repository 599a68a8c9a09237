// LDS-fed fp32 MFMA probe: each step reads its A/B fragments (4 x ds_read_b128) from LDS and runs 8 dependent
// v_mfma_f32_32x32x2_f32.  Variant A waits for the reads, then issues the MFMAs (what a K-tile loop with a barrier
// does); variant B issues the reads of step t+1 before the MFMAs of step t (software pipelined).  (tuning aid)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int PIPE, int NACC>
__global__ __launch_bounds__(256) void klds(float* out, const float* in, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 64 * 20 * 2];
    for (int i = threadIdx.x; i < 2 * 64 * 20 * 2; i += 256) sm[i] = in[i & 255];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, h = lane >> 5;
    const float* sa = sm + ((wave >> 1) * 32 + r) * 20 + 4 * h;
    const float* sb = sm + 64 * 20 + ((wave & 1) * 32 + r) * 20 + 4 * h;
    floatx16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] = 0.f;
    float4 a0, a1, b0, b1;
    if (PIPE) {
        a0 = *(const float4*)(sa); b0 = *(const float4*)(sb); a1 = *(const float4*)(sa + 8); b1 = *(const float4*)(sb + 8);
    }
    for (int it = 0; it < iters; ++it) {
        const int buf = (it & 1) * 2 * 64 * 20;
        float4 c0, c1, d0, d1;
        if (PIPE) {
            c0 = a0; c1 = a1; d0 = b0; d1 = b1;
            const int nb = ((it + 1) & 1) * 2 * 64 * 20;
            a0 = *(const float4*)(sa + nb); b0 = *(const float4*)(sb + nb); a1 = *(const float4*)(sa + nb + 8); b1 = *(const float4*)(sb + nb + 8);
        } else {
            c0 = *(const float4*)(sa + buf); d0 = *(const float4*)(sb + buf); c1 = *(const float4*)(sa + buf + 8); d1 = *(const float4*)(sb + buf + 8);
        }
#pragma unroll
        for (int a = 0; a < NACC; ++a) {
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.x, d0.x, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.y, d0.y, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.z, d0.z, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c0.w, d0.w, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.x, d1.x, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.y, d1.y, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.z, d1.z, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(c1.w, d1.w, acc[a], 0, 0, 0);
        }
        if ((it & 63) == 63) for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) acc[a][i] *= 1e-3f;
    }
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) for (int i = 0; i < 16; ++i) s += acc[a][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K kern, int nacc, int waves_per_simd, float* out, float* in) {
    const int iters = 2000;
    dim3 grid(256 * waves_per_simd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, in, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(kern, grid, block, 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 5.0 * 256 * waves_per_simd * 4 * (double)iters * 8 * nacc * 4096;
    printf("%s nacc=%d waves/SIMD=%d : %.1f TFLOP/s\n", name, nacc, waves_per_simd, flops / (ms * 1e-3) / 1e12);
}
int main() {
    float *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 1024);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    for (int w : {1, 2, 4, 6}) {
        run("wait-then-mfma", klds<0, 1>, 1, w, out, in);
        run("pipelined     ", klds<1, 1>, 1, w, out, in);
        run("wait-then-mfma", klds<0, 2>, 2, w, out, in);
        run("pipelined     ", klds<1, 2>, 2, w, out, in);
    }
    return 0;
}
