// Experiment harness: does the streaming rate of a FRESH process depend on how long / how often the box has been used?
// One process = allocate 1.5 GB, read 1 GB with nt loads REPS times, print the rate; optionally repeat alloc/free cycles
// inside the process (argv[1] = cycles, argv[2] = reps per cycle).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef double dvec2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_read(const dvec2* __restrict__ p, size_t n2, double* __restrict__ out)
{
    double acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; k + 7 * stride < n2; k += 8 * stride) {
        dvec2 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = __builtin_nontemporal_load(p + k + u * stride);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 12345.678) out[blockIdx.x] = acc;
}
int main(int argc, char** argv)
{
    const int cycles = argc > 1 ? atoi(argv[1]) : 1, reps = argc > 2 ? atoi(argv[2]) : 20;
    const size_t bytes = (size_t)1 << 30;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int c = 0; c < cycles; ++c) {
        char* buf; double* scratch;
        CK(hipMalloc(&buf, bytes + (512 << 20))); CK(hipMalloc(&scratch, 1 << 20));
        CK(hipMemset(buf, 1, bytes));
        for (int r = 0; r < 2; ++r) k_read<<<4096, 256>>>((const dvec2*)buf, bytes / 16, scratch);
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) k_read<<<4096, 256>>>((const dvec2*)buf, bytes / 16, scratch);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("cycle %d: %.0f GB/s\n", c, bytes * (double)reps / (ms * 1e-3) * 1e-9); fflush(stdout);
        CK(hipFree(buf)); CK(hipFree(scratch));
    }
    return 0;
}
