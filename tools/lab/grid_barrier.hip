// What does a device-wide barrier cost inside one launch?  The price of replacing the two launch boundaries of a
// DialogueRNN step by grid barriers in a persistent kernel (configuration 5).  NWG workgroups of 256
// threads, one per CU (96 KB of LDS each, as the persistent recurrence would hold its weight slice), run NB barriers:
// every workgroup's lane 0 draws a ticket (agent-scope atomic add); the last arriver bumps the generation word; the others
// poll it with relaxed agent-scope loads.  Between barriers every thread does a token amount of work on global memory
// that the NEXT phase of another workgroup reads (so the barrier has to publish data: release before the ticket, acquire
// after the poll).  EVERY spin is bounded: a workgroup that polls 4 M times without seeing the generation move sets the
// error word and leaves — the launch always drains.  Prints microseconds per barrier.
//   hipcc -O3 --offload-arch=gfx950 tools/lab/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ __launch_bounds__(256) void barrier_kernel(unsigned* count, unsigned* gen, unsigned* err, float* buf, int nwg, int nb) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wg = blockIdx.x;
    lds[tid] = (float)wg;
    __shared__ unsigned s_ok;
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) {
        // phase work: write my slot, later read a neighbour's slot of the previous phase
        buf[(size_t)(b & 1) * nwg * 256 + wg * 256 + tid] = acc + 1.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned target = (unsigned)b + 1u;
            const unsigned t = __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned ok = 1u;
            if (t == (unsigned)nwg * target - 1u) {
                __hip_atomic_store(gen, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // last arriver: open the gate
            } else {
                unsigned polls = 0;
                while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++polls > (4u << 20)) { ok = 0u; __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            s_ok = ok;
        }
        __syncthreads();
        if (!s_ok) return;                       // bounded: leave instead of hanging
        acc += buf[(size_t)(b & 1) * nwg * 256 + ((wg + 1) % nwg) * 256 + tid];
    }
    buf[(size_t)2 * nwg * 256 + wg * 256 + tid] = acc + lds[tid];
}

int main() {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    unsigned *count, *gen, *err;
    float* buf;
    hipMalloc(&count, 4); hipMalloc(&gen, 4); hipMalloc(&err, 4);
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute((const void*)barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nwg : {60, 126, 252}) {
        if (nwg > cus) continue;                 // one workgroup per CU (96 KB of LDS): all of them resident
        hipMalloc(&buf, (size_t)3 * nwg * 256 * sizeof(float));
        for (int nb : {1, 201, 401}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(count, 0, 4); hipMemset(gen, 0, 4); hipMemset(err, 0, 4);
                hipMemset(buf, 0, (size_t)3 * nwg * 256 * sizeof(float));
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(barrier_kernel, dim3(nwg), dim3(256), lds, 0, count, gen, err, buf, nwg, nb);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                unsigned herr = 0;
                hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
                if (herr) { printf("nwg %d nb %d: a spin ran out (error word set) — not all workgroups were resident?\n", nwg, nb); break; }
                if (ms < best) best = ms;
            }
            printf("workgroups %3d, barriers %3d: %8.2f us per launch\n", nwg, nb, best * 1e3f);
        }
        hipFree(buf);
    }
    printf("(per barrier = (t[401] - t[201]) / 200)\n");
    return 0;
}
