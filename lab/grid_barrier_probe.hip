// What a grid-wide barrier costs on this part: 256 workgroups (one per CU) of 256 threads meet NB times at a monotonic counter
// in device memory (agent-scope relaxed atomics: performed beyond the XCDs' L2s), optionally exchanging data the way a layer
// of a persistent few-row step kernel would (every workgroup stores 8 KiB with agent-scope stores, waits for them, arrives;
// after the barrier it loads 16 KiB that other workgroups wrote).  Every wait has a poll limit: a workgroup that runs into it
// poisons the counter so that every later wait passes at once, and the kernel ends.
// hipcc -O3 --offload-arch=gfx950 lab/grid_barrier_probe.hip -o lab/grid_barrier_probe && lab/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr unsigned kPoison = 0x40000000u;
constexpr int kPollLimit = 400000;

__device__ __forceinline__ void barrier(unsigned* cnt, unsigned target, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++polls > kPollLimit) {
                __hip_atomic_fetch_add(cnt, kPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *err = 1;
                break;
            }
        }
    }
    __syncthreads();
}

// two-level form: the workgroups of a group of 32 meet at their own counter, the group's last arrival goes to the top counter,
// everybody polls the top counter
__device__ __forceinline__ void barrier2(unsigned* cnt, unsigned* grp, unsigned round, unsigned ngrp, int* err) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned g = blockIdx.x & 7;   // consecutive workgroups go to consecutive XCDs: one group per XCD
        const unsigned members = gridDim.x / 8;
        const unsigned old = __hip_atomic_fetch_add(grp + g * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == round * members + members - 1) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int polls = 0;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (round + 1) * ngrp) {
            if (++polls > kPollLimit) {
                __hip_atomic_fetch_add(cnt, kPoison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *err = 1;
                break;
            }
        }
    }
    __syncthreads();
}

template <int MODE>   // 0: flat barrier only, 1: two-level barrier only, 2: flat + exchange, 3: two-level + exchange
__global__ __launch_bounds__(256) void probe(unsigned* cnt, unsigned* grp, float* buf, int nb, int* err, float* sink) {
    const int G = gridDim.x;
    float acc = 0.f;
    for (int b = 0; b < nb; ++b) {
        if (MODE >= 2) {
            float* mine = buf + ((size_t)(b & 1) * G + blockIdx.x) * 2048;
            for (int i = 0; i < 8; ++i) __hip_atomic_store(mine + i * 256 + threadIdx.x, (float)(b + i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (MODE & 1) barrier2(cnt, grp, (unsigned)b, 8, err); else barrier(cnt, (unsigned)(b + 1) * G, err);
        if (MODE >= 2) {
            // 16 KiB from two other workgroups (on other XCDs)
            for (int q = 0; q < 2; ++q) {
                const float* theirs = buf + ((size_t)(b & 1) * G + (blockIdx.x + 1 + q * 3) % G) * 2048;
                for (int i = 0; i < 8; ++i) acc += __hip_atomic_load(theirs + i * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (acc == -1.f) *sink = acc;
}

int main() {
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int G = cus < 256 ? cus & ~7 : 256;
    unsigned *cnt, *grp; float *buf, *sink; int* err;
    CHECK(hipMalloc(&cnt, 256)); CHECK(hipMalloc(&grp, 8 * 32 * 4)); CHECK(hipMalloc(&buf, (size_t)2 * G * 2048 * 4));
    CHECK(hipMalloc(&sink, 4)); CHECK(hipMalloc(&err, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int NB = 2000;
    for (int mode = 0; mode < 4; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipMemset(cnt, 0, 256)); CHECK(hipMemset(grp, 0, 8 * 32 * 4)); CHECK(hipMemset(err, 0, 4));
            CHECK(hipEventRecord(e0));
            switch (mode) {
                case 0: probe<0><<<G, 256>>>(cnt, grp, buf, NB, err, sink); break;
                case 1: probe<1><<<G, 256>>>(cnt, grp, buf, NB, err, sink); break;
                case 2: probe<2><<<G, 256>>>(cnt, grp, buf, NB, err, sink); break;
                default: probe<3><<<G, 256>>>(cnt, grp, buf, NB, err, sink); break;
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
            int herr = 0; CHECK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            printf("mode %d (%s%s) %d workgroups: %.2f us per round%s\n", mode, (mode & 1) ? "two-level" : "flat", mode >= 2 ? " + exchange" : "", G,
                   ms * 1e3 / NB, herr ? "  POLL LIMIT HIT" : "");
            if (herr) return 2;
        }
    }
    return 0;
}
