// Does VALU work issued between the MFMAs of ONE wave's dependent chain run in the MFMA's shadow?  One accumulator,
// fresh operand registers per MFMA, NV independent VALU instructions (xor chain or v_mad_u64_u32) pinned after every
// MFMA with sched_barrier; 1 / 2 / 3 waves per SIMD.  Prints ns per MFMA slot (64 cycles = 26.7 ns at 2.4 GHz).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float floatx16 __attribute__((ext_vector_type(16)));

template <int NV, int KIND>   // KIND 0: v_xor/v_add mix, 1: v_mad_u64_u32, 2: global store every slot
__global__ void kern(float* out, const float* in, int iters, float* sink) {
    floatx16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float x[8], y[8];
    for (int u = 0; u < 8; ++u) { x[u] = in[(threadIdx.x * 7 + u * 13) & 255]; y[u] = in[(threadIdx.x * 5 + u * 29 + 3) & 255]; }
    uint32_t v0 = threadIdx.x, v1 = threadIdx.x * 3 + 1;
    float* sp = sink + (size_t)(blockIdx.x * blockDim.x + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x[u], y[u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if (KIND == 1) {
                    uint64_t p, cy;
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(p), "=s"(cy) : "s"(0xD2511F53u), "v"(v0));
                    v0 = (uint32_t)(p >> 32) ^ v1; v1 = (uint32_t)p;
                } else {
                    asm volatile("v_xor_b32 %0, %1, %2" : "=v"(v0) : "v"(v0), "v"(v1));
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(v1) : "v"(v0), "v"(v1));
                }
            }
            if (KIND == 2) sp[(size_t)(u & 1) * 1048576] = __uint_as_float(v0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((it & 63) == 63) for (int i = 0; i < 16; ++i) acc[i] *= 1e-3f;
    }
    float s = __uint_as_float(v0 ^ v1);
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename K>
void run(const char* name, K k, int nv, int w, float* out, float* in, float* sink) {
    const int iters = 1000;
    dim3 grid(256 * w), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, grid, block, 0, 0, out, in, 20, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, grid, block, 0, 0, out, in, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double slots_per_simd = 3.0 * iters * 8 * w;       // MFMAs through one SIMD's pipe
    printf("%-28s NV=%2d waves/SIMD=%d : %6.1f ns per MFMA of the pipe  (%5.1f ns per slot of a wave)\n", name, nv, w,
           ms * 1e6 / slots_per_simd, ms * 1e6 / (3.0 * iters * 8));
}
int main() {
    float *out, *in, *sink; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 1024); hipMalloc(&sink, (size_t)1048576 * 4 * 4);
    float h[256]; for (int i = 0; i < 256; ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    hipMemcpy(in, h, 1024, hipMemcpyHostToDevice);
    for (int w : {1, 2, 3}) {
        run("mfma only", kern<0, 0>, 0, w, out, in, sink);
        run("+ xor/add pairs", kern<2, 0>, 4, w, out, in, sink);
        run("+ xor/add pairs", kern<4, 0>, 8, w, out, in, sink);
        run("+ xor/add pairs", kern<6, 0>, 12, w, out, in, sink);
        run("+ xor/add pairs", kern<8, 0>, 16, w, out, in, sink);
        run("+ xor/add pairs", kern<12, 0>, 24, w, out, in, sink);
        run("+ mad_u64 (+2 valu)", kern<1, 1>, 1, w, out, in, sink);
        run("+ mad_u64 (+2 valu)", kern<2, 1>, 2, w, out, in, sink);
        run("+ mad_u64 (+2 valu)", kern<4, 1>, 4, w, out, in, sink);
        run("+ 4 valu + dword store", kern<2, 2>, 4, w, out, in, sink);
    }
    return 0;
}
