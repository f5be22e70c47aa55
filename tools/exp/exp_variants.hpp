// kernel variants under test (edit freely; exp_tile.hip times each against the baseline on the same box)
#pragma once
namespace mi {

// V1: staging only (row loop skipped): how long do the loads of a tile take by themselves?
template <int BS>
__global__ __launch_bounds__(BS) void k_stage_only(const TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x, per = gridDim.x >> 3;
    const int t = (b < (per << 3)) ? (b & 7) * per + (b >> 3) : b;
    double* cU = smem; double* xs = smem + a.offX;
    const int tid = threadIdx.x;
    const int c0 = a.tileCellStart[t], nc = a.tileCellStart[t + 1] - c0;
    const int s0 = a.tileSlotStart[t], ns = a.tileSlotStart[t + 1] - s0;
    const int h0 = a.tileHaloStart[t], nh = a.tileHaloStart[t + 1] - h0;
    stage_copy_nt<BS>(reinterpret_cast<const double2*>(a.up + s0), reinterpret_cast<double2*>(cU), ns >> 1, tid, false);
    stage_copy<BS>(a.x + c0, xs, nc, tid);
    stage_gather<BS>(a.x, a.haloCell + h0, xs + nc, nh, tid);
    __syncthreads();
    for (int i = tid; i < nc; i += BS) a.y[c0 + i] = xs[i] + cU[i];
}

// V2: staging + entry stream, no LDS gathers (entries are read and folded into the result)
template <int BS>
__global__ __launch_bounds__(BS) void k_stage_entries(const TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = blockIdx.x, per = gridDim.x >> 3;
    const int t = (b < (per << 3)) ? (b & 7) * per + (b >> 3) : b;
    double* cU = smem; double* xs = smem + a.offX;
    const int tid = threadIdx.x;
    const int c0 = a.tileCellStart[t], nc = a.tileCellStart[t + 1] - c0;
    const int s0 = a.tileSlotStart[t], ns = a.tileSlotStart[t + 1] - s0;
    const int h0 = a.tileHaloStart[t], nh = a.tileHaloStart[t + 1] - h0;
    stage_copy_nt<BS>(reinterpret_cast<const double2*>(a.up + s0), reinterpret_cast<double2*>(cU), ns >> 1, tid, false);
    stage_copy<BS>(a.x + c0, xs, nc, tid);
    stage_gather<BS>(a.x, a.haloCell + h0, xs + nc, nh, tid);
    const int sl0 = a.tileSliceStart[t];
    const int e0 = a.sliceEntryStart[sl0], e1 = a.sliceEntryStart[a.tileSliceStart[t + 1]];
    uint32_t acc = 0;
    for (int k = e0 + tid; k < e1; k += BS) acc ^= a.entries[k];
    __syncthreads();
    for (int i = tid; i < nc; i += BS) a.y[c0 + i] = xs[i] + cU[i] + (double)(acc & 1u);
}

// calibration: plain streaming read of nBytes (16-byte loads, UNR in flight per thread), one partial per block
template <int UNR>
__global__ __launch_bounds__(256) void k_stream_read(const double2* __restrict__ p, size_t n2, double* __restrict__ out)
{
    double acc = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; k + (UNR - 1) * stride < n2; k += UNR * stride) {
        double2 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = p[k + u * stride];
#pragma unroll
        for (int u = 0; u < UNR; ++u) acc += v[u].x + v[u].y;
    }
    for (; k < n2; k += stride) acc += p[k].x + p[k].y;
    if (acc == 12345.678) out[blockIdx.x] = acc;
}
// calibration: streaming copy y = x (8-byte elements as double2)
__global__ __launch_bounds__(256) void k_stream_copy(const double2* __restrict__ p, double2* __restrict__ q, size_t n2)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t k = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; k + 3 * stride < n2; k += 4 * stride) {
        const double2 a = p[k], b = p[k + stride], c = p[k + 2 * stride], d = p[k + 3 * stride];
        q[k] = a; q[k + stride] = b; q[k + 2 * stride] = c; q[k + 3 * stride] = d;
    }
    for (; k < n2; k += stride) q[k] = p[k];
}

template <class Run, class Check>
void exp_variants(TileArgs& a, const TileLayout& L, size_t lds, Run run, Check check, double* y1)
{
    static bool once = false;
    if (!once) {
        once = true;
        (void)hipFuncSetAttribute((const void*)k_stage_only<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        (void)hipFuncSetAttribute((const void*)k_stage_entries<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        (void)hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
    }
    {   // calibration lines print their own bandwidth (the harness's GB/s column assumes Amul's algorithmic bytes)
        const size_t n2 = (size_t)L.totalSlots / 2;
        static double* scratch = nullptr; if (!scratch) (void)hipMalloc(&scratch, sizeof(double) * 65536);
        printf("   [calibration: the next lines move %.1f MB (read) / %.1f MB (copy r+w); Amul's algorithmic bytes are %.1f MB]\n", n2 * 16e-6, 2 * (L.nCells / 2) * 16e-6, (24.0 * L.nCells + 16.0 * L.nFaces) * 1e-6);
        run("stream read x4 grid 2048", [&] { k_stream_read<4><<<2048, 256, 0, 0>>>(reinterpret_cast<const double2*>(a.up), n2, scratch); }, y1);
        run("stream read x8 grid 4096", [&] { k_stream_read<8><<<4096, 256, 0, 0>>>(reinterpret_cast<const double2*>(a.up), n2, scratch); }, y1);
        run("stream read x8 grid 16384", [&] { k_stream_read<8><<<16384, 256, 0, 0>>>(reinterpret_cast<const double2*>(a.up), n2, scratch); }, y1);
        run("stream copy x->y grid 4096", [&] { k_stream_copy<<<4096, 256, 0, 0>>>(reinterpret_cast<const double2*>(a.x), reinterpret_cast<double2*>(y1), (size_t)L.nCells / 2); }, y1);
    }
    run("stage only", [&] { k_stage_only<512><<<L.nTiles, 512, lds, 0>>>(a); }, y1);
    run("stage + entry stream", [&] { k_stage_entries<512><<<L.nTiles, 512, lds, 0>>>(a); }, y1);
    if (L.compact) { run("compact entries", [&] { tile_kernel<OP_AMUL, false, false, 512, true><<<L.nTiles, 512, lds, 0>>>(a); }, y1); check("compact"); }
}
} // namespace mi
