// Experiment harness (not a product path): what the memory side of an MI355X gives to streams that are RE-READ every
// solver iteration (the matrix: coefficients, row entries, diagonal) next to streams that are not (the vectors).
//   1. re-read bandwidth of one buffer of S MB, default policy vs nt loads: where does the Infinity Cache (256 MiB) end?
//   2. a resident candidate A (default policy) interleaved with a large stream B (default / nt): does A stay on-die?
//   3. mixed read/write streams as the PCG vector kernels issue them (3 reads + 1 write, 5 reads + 2 writes).
// Build: make -C tools/exp exp_mall ; run: tools/exp/exp_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

typedef double dvec2 __attribute__((ext_vector_type(2)));

template <int UNR, bool NT>
__global__ __launch_bounds__(256) void k_read(const dvec2* __restrict__ p, size_t n2, double* __restrict__ out)
{
    double acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; k + (UNR - 1) * stride < n2; k += UNR * stride) {
        dvec2 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = NT ? __builtin_nontemporal_load(p + k + u * stride) : p[k + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y;
    }
    for (; k < n2; k += stride) acc += p[k].x + p[k].y;
    if (acc == 12345.678) out[blockIdx.x] = acc;
}

// block-contiguous chunks (the engine's vector kernels): NR inputs, NW outputs; out_w = sum of inputs (* small factor)
template <int NR, int NWR, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k_rw(const dvec2* const* __restrict__ in, dvec2* const* __restrict__ outp, size_t n2)
{
    const size_t chunk = (n2 + gridDim.x - 1) / gridDim.x;
    const size_t lo = (size_t)blockIdx.x * chunk, hi = lo + chunk < n2 ? lo + chunk : n2;
    const dvec2* src[NR]; dvec2* dst[NWR];
#pragma unroll
    for (int r = 0; r < NR; ++r) src[r] = in[r];
#pragma unroll
    for (int w = 0; w < NWR; ++w) dst[w] = outp[w];
#pragma unroll 4
    for (size_t k = lo + threadIdx.x; k < hi; k += 256) {
        dvec2 s = {0.0, 0.0};
#pragma unroll
        for (int r = 0; r < NR; ++r) { const dvec2 v = NTL ? __builtin_nontemporal_load(src[r] + k) : src[r][k]; s += v; }
#pragma unroll
        for (int w = 0; w < NWR; ++w) { const dvec2 o = s * (1.0 + w); if (NTS) __builtin_nontemporal_store(o, dst[w] + k); else dst[w][k] = o; }
    }
}

int main()
{
    const size_t MB = 1 << 20;
    const size_t cap = 1536 * MB;
    char* buf; CK(hipMalloc(&buf, cap)); CK(hipMemset(buf, 1, cap));
    double* scratch; CK(hipMalloc(&scratch, 1 << 20));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, int reps) {
        for (int r = 0; r < 2; ++r) launch();
        CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return (double)ms * 1e3 / reps; // us per launch
    };
    printf("== 1. one buffer of S MB re-read back to back (grid 4096 x 256, 8 x 16-byte loads in flight per lane)\n");
    for (size_t S : {32, 64, 128, 192, 224, 256, 320, 384, 512, 768, 1024}) {
        const size_t n2 = S * MB / 16;
        const double us0 = timeit([&] { k_read<8, false><<<4096, 256>>>((const dvec2*)buf, n2, scratch); }, 20);
        const double us1 = timeit([&] { k_read<8, true><<<4096, 256>>>((const dvec2*)buf, n2, scratch); }, 20);
        printf("S %5zu MB   default %8.1f us %7.0f GB/s     nt %8.1f us %7.0f GB/s\n", S, us0, S * MB / us0 * 1e-3, us1, S * MB / us1 * 1e-3);
    }
    printf("== 2. resident candidate A (default policy) + stream B of 640 MB per round; time of the A read alone inside the round\n");
    for (size_t SA : {64, 128, 192, 224}) {
        for (int ntB = 0; ntB < 2; ++ntB) {
            const size_t nA = SA * MB / 16, nB = 640 * MB / 16;
            const dvec2* A = (const dvec2*)buf; const dvec2* B = (const dvec2*)(buf + 512 * MB);
            // warm
            for (int r = 0; r < 3; ++r) { k_read<8, false><<<4096, 256>>>(A, nA, scratch); if (ntB) k_read<8, true><<<4096, 256>>>(B, nB, scratch); else k_read<8, false><<<4096, 256>>>(B, nB, scratch); }
            CK(hipDeviceSynchronize());
            double tA = 0, tB = 0; const int R = 10;
            hipEvent_t a0, a1, b1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1)); CK(hipEventCreate(&b1));
            for (int r = 0; r < R; ++r) {
                CK(hipEventRecord(a0));
                k_read<8, false><<<4096, 256>>>(A, nA, scratch);
                CK(hipEventRecord(a1));
                if (ntB) k_read<8, true><<<4096, 256>>>(B, nB, scratch); else k_read<8, false><<<4096, 256>>>(B, nB, scratch);
                CK(hipEventRecord(b1)); CK(hipEventSynchronize(b1));
                float m; CK(hipEventElapsedTime(&m, a0, a1)); tA += m; CK(hipEventElapsedTime(&m, a1, b1)); tB += m;
            }
            printf("A %4zu MB (default) + B 640 MB (%s):  A %7.1f us %7.0f GB/s    B %7.1f us %7.0f GB/s\n", SA, ntB ? "nt     " : "default",
                   tA * 1e3 / R, SA * MB / (tA * 1e3 / R) * 1e-3, tB * 1e3 / R, 640.0 * MB / (tB * 1e3 / R) * 1e-3);
        }
    }
    printf("== 3. vector-kernel shaped streams over 80.6 MB vectors (10 077 696 doubles), grid x 256 threads, block-contiguous chunks\n");
    {
        const size_t n2 = 10077696 / 2;
        std::vector<const dvec2*> hin; std::vector<dvec2*> hout;
        for (int r = 0; r < 6; ++r) hin.push_back((const dvec2*)(buf + (size_t)r * 96 * MB));
        for (int w = 0; w < 2; ++w) hout.push_back((dvec2*)(buf + (size_t)(8 + w) * 96 * MB));
        const dvec2** din; dvec2** dout; CK(hipMalloc(&din, 64)); CK(hipMalloc(&dout, 64));
        CK(hipMemcpy(din, hin.data(), sizeof(void*) * hin.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dout, hout.data(), sizeof(void*) * hout.size(), hipMemcpyHostToDevice));
        for (int grid : {1024, 2048, 4096, 8192}) {
            const double a = timeit([&] { k_rw<3, 1, false, false><<<grid, 256>>>(din, dout, n2); }, 20);
            const double b = timeit([&] { k_rw<3, 1, true, true><<<grid, 256>>>(din, dout, n2); }, 20);
            const double c = timeit([&] { k_rw<5, 2, false, false><<<grid, 256>>>(din, dout, n2); }, 20);
            const double d = timeit([&] { k_rw<5, 2, true, true><<<grid, 256>>>(din, dout, n2); }, 20);
            const double e = timeit([&] { k_rw<1, 1, false, false><<<grid, 256>>>(din, dout, n2); }, 20);
            const double bytes = 10077696.0 * 8;
            printf("grid %5d   3r1w %6.1f us %5.0f GB/s (nt %6.1f us %5.0f)   5r2w %6.1f us %5.0f GB/s (nt %6.1f us %5.0f)   1r1w %6.1f us %5.0f GB/s\n", grid,
                   a, 4 * bytes / a * 1e-3, b, 4 * bytes / b * 1e-3, c, 7 * bytes / c * 1e-3, d, 7 * bytes / d * 1e-3, e, 2 * bytes / e * 1e-3);
        }
    }
    return 0;
}
