// How fast can a kernel's workgroups store x' into pinned host memory?  (the scaler epilogue of the fused sweep: 6496
// walkers x 32 doubles = 1.66 MB per launch.)  Measurement only.
//   pattern F: column-major [D][n] -- a workgroup of 16 walkers stores 128 contiguous bytes per column (what the step does)
//   pattern C: row-major [n][D]    -- a workgroup stores 4 KB contiguous
// Time = launch -> the host sees the completion word (written by the last workgroup behind a system fence), minus the
// same for a kernel that stores nothing but the word.
// build: hipcc --offload-arch=gfx950 -O3 pcie_store.hip -o pcie_store
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

// FENCE 0: __threadfence_system() in every workgroup (writes back L2: what the step did until round 3); 1: agent-scope
// release in every workgroup (a wait for the stores' acknowledgements), the completion word a system-scope release store by
// the last workgroup (which writes L2 back once); 2: as 1, the word a relaxed system-scope store behind an agent-scope
// release (no write-back at all: everything the host reads behind the word is in memory the device does not cache)
template <int FENCE>
__global__ __launch_bounds__(128) void k(double* host, long long n, int D, int pattern, unsigned* ticket, long long* flag,
                                          long long value, int spin) {
    const int tid = threadIdx.x;
    const long long row0 = (long long)blockIdx.x * 16;
    if (spin) { long long t0 = clock64(); while (clock64() - t0 < spin) { } }       // (stands for the sweep: everyone finishes together)
    if (pattern >= 0) {
        for (int e = tid; e < 16 * D; e += 128) {
            const int r = e & 15, j = e >> 4;
            if (row0 + r < n) {
                const double v = (double)(row0 + r) + 1e-3 * j + 1e4 * (double)value;
                if (pattern == 0) host[(long long)j * n + row0 + r] = v;
                else host[(row0 + r) * D + j] = v;
            }
        }
    }
    if (FENCE == 0) __threadfence_system();
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (t == gridDim.x - 1) {
            *ticket = 0;
            if (FENCE == 2) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

int main(int argc, char** argv) {
    const long long n = argc > 1 ? atoll(argv[1]) : 6496;
    const int D = 32;
    double* host; hipHostMalloc(&host, n * D * sizeof(double), hipHostMallocDefault);
    long long* flag; hipHostMalloc(&flag, 64, hipHostMallocDefault);
    unsigned* ticket; hipMalloc(&ticket, 4); hipMemset(ticket, 0, 4);
    *flag = 0;
    const int grid = (int)((n + 15) / 16);
    long long val = 0;
    for (int fence : {0, 1, 2})
    for (int spin : {0, 100000}) {
        double base = 0;
        for (int pattern : {-1, 0}) {
            double best = 1e9, sum = 0; const int reps = 2000;
            long long bad = 0;
            for (int it = 0; it < reps + 20; ++it) {
                ++val;
                hipDeviceSynchronize();
                auto t0 = std::chrono::steady_clock::now();
                if (fence == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(128), 0, 0, host, n, D, pattern, ticket, flag, val, spin);
                else if (fence == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(128), 0, 0, host, n, D, pattern, ticket, flag, val, spin);
                else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(128), 0, 0, host, n, D, pattern, ticket, flag, val, spin);
                while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != val) { }
                auto t1 = std::chrono::steady_clock::now();
                // the word is there: is every x' of THIS launch there?  (read at once, before the kernel has ended)
                if (pattern == 0)
                    for (long long e = 0; e < n * D; e += 1) {
                        const long long j = e / n, r = e - j * n;
                        const double want = (double)r + 1e-3 * j + 1e4 * (double)val;
                        if (((volatile double*)host)[e] != want) ++bad;
                    }
                const double us = std::chrono::duration<double, std::micro>(t1 - t0).count();
                if (it >= 20) { sum += us; if (us < best) best = us; }
            }
            if (pattern < 0) base = sum / reps;
            printf("fence %s  spin %6d cycles  %-22s launch -> flag: mean %6.1f us  min %6.1f us   (minus the empty kernel: %5.1f us -> %5.1f GB/s)  stale words seen: %lld\n",
                   fence == 0 ? "system       " : fence == 1 ? "agent        " : "agent,relaxed", spin, pattern < 0 ? "no stores" : "F: 128-byte chunks", sum / reps, best,
                   sum / reps - base, pattern < 0 ? 0.0 : n * D * 8 / (sum / reps - base) * 1e-3, bad);
        }
    }
    return 0;
}
