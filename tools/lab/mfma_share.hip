// fp32 MFMA issue rate for the OPERAND-SHARING patterns of the d_model-100 kernels (round 5): does a 32x32x2 tiling of the
// 100-wide dimension (3 tiles of 32 sharing the other operand, + a 4-row tail) issue faster than today's 16x16x4 tiling
// (6 tiles of 16 sharing it)?  Register operands, random data, 1..3 waves per SIMD; TFLOP/s over ALL issued MFMA flops.
//   p16x7 : 16x16x4, k-step = 7 MFMAs with 7 different A registers and ONE B register (tn100 / n100 today, padded form)
//   p16x6t: 16x16x4 x 6 + one 4x4x1 (today's TAIL4 form)
//   p32x3 : 32x32x2, k-step = 3 MFMAs with 3 different A registers and one B register
//   p32x3t: 32x32x2 x 3 + one 4x4x1 tail
//   p16same / p32same: every MFMA the same two registers (the ceiling)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int KU = 8;   // k-steps per loop iteration (distinct register sets)

template <int MODE>
__global__ __launch_bounds__(256) void kern(float* out, const float* in, int iters) {
    float a[KU][7], b[KU];
    for (int u = 0; u < KU; ++u) {
        b[u] = in[(threadIdx.x * 5 + u * 29 + 3) & 255];
        for (int m = 0; m < 7; ++m) a[u][m] = in[(threadIdx.x * 7 + u * 13 + m * 31) & 255];
    }
    floatx4 c4[7];
    floatx16 c16[4];
    for (int m = 0; m < 7; ++m) c4[m] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < 4; ++m) for (int i = 0; i < 16; ++i) c16[m][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            if (MODE == 0) {          // p16x7
#pragma unroll
                for (int m = 0; m < 7; ++m) c4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][m], b[u], c4[m], 0, 0, 0);
            } else if (MODE == 1) {   // p16x6t
#pragma unroll
                for (int m = 0; m < 6; ++m) c4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][m], b[u], c4[m], 0, 0, 0);
                c4[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][6], b[u], c4[6], 0, 0, 0);
            } else if (MODE == 2) {   // p32x3
#pragma unroll
                for (int m = 0; m < 3; ++m) c16[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u], c16[m], 0, 0, 0);
            } else if (MODE == 3) {   // p32x3t
#pragma unroll
                for (int m = 0; m < 3; ++m) c16[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], b[u], c16[m], 0, 0, 0);
                c4[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(a[u][6], b[u], c4[6], 0, 0, 0);
            } else if (MODE == 4) {   // p16same
#pragma unroll
                for (int m = 0; m < 7; ++m) c4[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0][0], b[0], c4[m], 0, 0, 0);
            } else if (MODE == 5) {   // p32same
#pragma unroll
                for (int m = 0; m < 3; ++m) c16[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][0], b[0], c16[m], 0, 0, 0);
            } else if (MODE == 6) {   // p32x1: ONE accumulator, new A and new B at every MFMA (gemm_body / gemm_wres_kernel today)
#pragma unroll
                for (int m = 0; m < 3; ++m) c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][m], a[u][m + 3], c16[0], 0, 0, 0);
            } else if (MODE == 7) {   // p32x1s: one accumulator, the same operands (the dependent chain alone)
#pragma unroll
                for (int m = 0; m < 3; ++m) c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0][0], b[0], c16[0], 0, 0, 0);
            } else if (MODE == 8) {   // p32x2: two accumulators sharing A, new A and new B pair per step (64 x 128 wave tiles)
                c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][0], a[u][1], c16[0], 0, 0, 0);
                c16[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][0], a[u][2], c16[1], 0, 0, 0);
                c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][3], a[u][4], c16[0], 0, 0, 0);
                c16[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][3], a[u][5], c16[1], 0, 0, 0);
            } else {                  // p32x4: 2 x 2 accumulators walked boustrophedon (one operand changes per MFMA)
                c16[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][0], a[u][2], c16[0], 0, 0, 0);
                c16[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][0], a[u][3], c16[1], 0, 0, 0);
                c16[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][1], a[u][3], c16[2], 0, 0, 0);
                c16[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][1], a[u][2], c16[3], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((it & 63) == 63) {
            for (int m = 0; m < 7; ++m) for (int i = 0; i < 4; ++i) c4[m][i] *= 1e-3f;
            for (int m = 0; m < 4; ++m) for (int i = 0; i < 16; ++i) c16[m][i] *= 1e-3f;
        }
    }
    float s = 0.f;
    for (int m = 0; m < 7; ++m) for (int i = 0; i < 4; ++i) s += c4[m][i];
    for (int m = 0; m < 4; ++m) for (int i = 0; i < 16; ++i) s += c16[m][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, double flop_per_kstep, double cyc_ideal, int wps, float* out, float* in) {
    const int iters = 1500;
    dim3 grid(256 * wps), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern<MODE>, grid, block, 0, 0, out, in, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(kern<MODE>, grid, block, 0, 0, out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ksteps = 5.0 * 256 * wps * 4 * (double)iters * KU;          // per-wave k-steps, all waves
    const double per_simd = ksteps / 1024.0;                                  // k-steps one SIMD issued
    printf("%-8s waves/SIMD=%d : %6.1f TFLOP/s issued | %.1f ns per k-step per SIMD = %.0f cycles at 2.4 GHz (MFMA passes alone: %.0f)\n", name, wps,
           ksteps * flop_per_kstep / (ms * 1e-3) / 1e12, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4, cyc_ideal);
}
int main() {
    float *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 1024);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    for (int w : {1, 2, 3}) {
        run<0>("p16x7", 7 * 2048.0, 7 * 32, w, out, in);
        run<1>("p16x6t", 6 * 2048.0 + 512.0, 6 * 32 + 8, w, out, in);
        run<2>("p32x3", 3 * 4096.0, 3 * 64, w, out, in);
        run<3>("p32x3t", 3 * 4096.0 + 512.0, 3 * 64 + 8, w, out, in);
        run<4>("p16same", 7 * 2048.0, 7 * 32, w, out, in);
        run<5>("p32same", 3 * 4096.0, 3 * 64, w, out, in);
        run<6>("p32x1", 3 * 4096.0, 3 * 64, w, out, in);
        run<7>("p32x1s", 3 * 4096.0, 3 * 64, w, out, in);
        run<8>("p32x2", 4 * 4096.0, 4 * 64, w, out, in);
        run<9>("p32x4", 4 * 4096.0, 4 * 64, w, out, in);
    }
    return 0;
}
