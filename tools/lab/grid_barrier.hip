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

// ---- XCD-hierarchical barrier (round 5; MI355X_MICROARCH.md "barrier-xcd"): a workgroup arrives at the counter of ITS XCD (the
// hardware XCC_ID, read with s_getreg: 8 XCDs, 32 CUs and one 4 MiB L2 each); the last arriver of an XCD — and only it — makes
// the XCD's stores visible device-wide (one release fence = one L2 write-back per XCD instead of one per workgroup), arrives at
// the top counter (8 arrivals instead of 252 on one word), waits for the top generation, acquires, and opens its XCD's gate; the
// other workgroups poll their XCD's generation word and acquire.  The number of workgroups per XCD is not known before the
// launch: a first flat barrier counts them.  Every phase's hand-off is CHECKED (the neighbour's slot must hold the phase
// number): a barrier that is fast because it publishes nothing would show up in `bad`.
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(v));
    return v & 7u;
}
struct XcdBar {
    unsigned size[8 * 16];      // workgroups on XCD x (at [16 x]: one 64-byte line per word)
    unsigned cnt[8 * 16];       // arrivals on XCD x, monotonic
    unsigned gen[8 * 16];       // generation of XCD x
    unsigned top[16], topgen[16];
    unsigned flat[16], flatgen[16];
    unsigned err[16], bad[16];
};
__device__ __forceinline__ bool spin_until(unsigned* w, unsigned target, unsigned* err) {
    unsigned polls = 0;
    while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++polls > (4u << 20)) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
    }
    return true;
}
__global__ __launch_bounds__(256) void barrier_xcd_kernel(XcdBar* B, float* buf, int nwg, int nb) {
    extern __shared__ float lds[];
    const int tid = threadIdx.x, wg = blockIdx.x;
    lds[tid] = (float)wg;
    __shared__ unsigned s_ok, s_x, s_n;
    // setup: count the workgroups of every XCD, one flat barrier
    if (tid == 0) {
        const unsigned x = xcc_id();
        __hip_atomic_fetch_add(&B->size[16 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const unsigned t = __hip_atomic_fetch_add(&B->flat[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool ok = true;
        if (t == (unsigned)nwg - 1u) __hip_atomic_store(&B->flatgen[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else ok = spin_until(&B->flatgen[0], 1u, &B->err[0]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        unsigned groups = 0;
        for (int i = 0; i < 8; ++i) groups += __hip_atomic_load(&B->size[16 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
        s_ok = ok ? 1u : 0u; s_x = x; s_n = groups;
    }
    __syncthreads();
    if (!s_ok) return;
    const unsigned x = s_x, groups = s_n;
    const unsigned mysize = __hip_atomic_load(&B->size[16 * x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned bad = 0;
    for (int b = 0; b < nb; ++b) {
        buf[(size_t)(b & 1) * nwg * 256 + wg * 256 + tid] = (float)(b + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (unsigned)b + 1u;
            bool ok = true;
            const unsigned t = __hip_atomic_fetch_add(&B->cnt[16 * x], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t == mysize * target - 1u) {
                // XCD leader: publish this XCD's stores (every workgroup of it drained its stores into the shared L2 before arriving)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned tt = __hip_atomic_fetch_add(&B->top[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (tt == groups * target - 1u) __hip_atomic_store(&B->topgen[0], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else ok = spin_until(&B->topgen[0], target, &B->err[0]);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(&B->gen[16 * x], target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                ok = spin_until(&B->gen[16 * x], target, &B->err[0]);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            s_ok = ok ? 1u : 0u;
        }
        __syncthreads();
        if (!s_ok) return;
        const float v = buf[(size_t)(b & 1) * nwg * 256 + ((wg + 1) % nwg) * 256 + tid];
        bad += (v != (float)(b + 1)) ? 1u : 0u;
    }
    if (bad) __hip_atomic_fetch_add(&B->bad[0], bad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    buf[(size_t)2 * nwg * 256 + wg * 256 + tid] = lds[tid];
}

static void run_xcd(int cus) {
    XcdBar* B; float* buf;
    hipMalloc(&B, sizeof(XcdBar));
    const size_t lds = 96 * 1024;
    hipFuncSetAttribute((const void*)barrier_xcd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int nwg : {60, 126, 252, 256}) {
        if (nwg > cus) continue;
        hipMalloc(&buf, (size_t)3 * nwg * 256 * sizeof(float));
        float tt[3] = {0, 0, 0};
        int i = 0;
        for (int nb : {1, 201, 401}) {
            float best = 1e30f;
            for (int rep = 0; rep < 5; ++rep) {
                hipMemset(B, 0, sizeof(XcdBar));
                hipMemset(buf, 0, (size_t)3 * nwg * 256 * sizeof(float));
                hipEvent_t e0, e1;
                hipEventCreate(&e0); hipEventCreate(&e1);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(barrier_xcd_kernel, dim3(nwg), dim3(256), lds, 0, B, buf, nwg, nb);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms = 0.f;
                hipEventElapsedTime(&ms, e0, e1);
                XcdBar h;
                hipMemcpy(&h, B, sizeof(XcdBar), hipMemcpyDeviceToHost);
                if (h.err[0]) { printf("xcd: nwg %d nb %d: a spin ran out (error word set)\n", nwg, nb); best = -1.f; break; }
                if (h.bad[0]) { printf("xcd: nwg %d nb %d: %u STALE hand-off reads\n", nwg, nb, h.bad[0]); }
                if (rep == 0 && nb == 1) {
                    printf("xcd: workgroups per XCD at %d:", nwg);
                    for (int x = 0; x < 8; ++x) printf(" %u", h.size[16 * x]);
                    printf("\n");
                }
                if (ms < best) best = ms;
            }
            tt[i++] = best;
            printf("xcd-hierarchical: workgroups %3d, barriers %3d: %8.2f us per launch\n", nwg, nb, best * 1e3f);
        }
        printf("xcd-hierarchical: workgroups %3d: %.2f us per barrier\n", nwg, (tt[2] - tt[1]) * 1e3f / 200.f);
        hipFree(buf);
    }
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
    run_xcd(cus);
    return 0;
}
