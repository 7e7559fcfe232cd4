// Dependent-launch boundary on one stream: N small kernels back to back, issued one by one vs replayed as a captured
// hipGraph (scripts/micro: measurement only, not part of the library).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void step_kernel(float* a, int n, int phase) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) a[i] = a[i] * 1.0001f + (float)phase;
}

int main(int argc, char** argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 68, blocks = argc > 2 ? atoi(argv[2]) : 256, reps = 20;
    float* a; CK(hipMalloc(&a, (size_t)blocks * 256 * 4)); CK(hipMemset(a, 0, (size_t)blocks * 256 * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    for (int w = 0; w < 3; ++w) for (int k = 0; k < N; ++k) step_kernel<<<blocks, 256, 0, s>>>(a, blocks * 256, k);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) for (int k = 0; k < N; ++k) step_kernel<<<blocks, 256, 0, s>>>(a, blocks * 256, k);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("stream launches: %d kernels x %d blocks: %.2f us per kernel\n", N, blocks, ms * 1e3 / (reps * N));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < N; ++k) step_kernel<<<blocks, 256, 0, s>>>(a, blocks * 256, k);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph replay:    %d kernels x %d blocks: %.2f us per kernel\n", N, blocks, ms * 1e3 / (reps * N));
    return 0;
}
