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

// ---------------------------------------------------------------------------------------------------------------------
// V3: persistent workgroups with REGISTER-STAGED PREFETCH of the next tile (single LDS image per workgroup).
//     While the rows of tile k are computed out of LDS, the global loads of tile k+1 (coefficients, psi, halo gather) are
//     already in flight into registers; they are committed to LDS between two barriers once tile k is done.
// ---------------------------------------------------------------------------------------------------------------------
namespace mi {
template <int BS>
__global__ __launch_bounds__(BS) void k_amul_prefetch(const TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* cU = smem;
    double* xs = smem + a.offX;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = BS / 64, PRE = 8, CU4 = 4, XV = 2, HV = 2;
    const int b = blockIdx.x, G = gridDim.x, nT = a.nPos;
    const int per = G >> 3, xc = b & 7, j = b >> 3;
    const int x0 = (int)((long long)xc * nT / 8), x1 = (int)((long long)(xc + 1) * nT / 8);
    const int p0 = x0 + (int)((long long)j * (x1 - x0) / per), p1 = x0 + (int)((long long)(j + 1) * (x1 - x0) / per);
    if (p0 >= p1) return;

    // registers that carry the NEXT tile
    double2 rc[CU4]; double rx[XV]; double rh[HV];
    int c0n, ncn, s0n, nsn, h0n, nhn, sl0n, nsln;
    auto prefetch = [&](int t) {
        c0n = a.tileCellStart[t]; ncn = a.tileCellStart[t + 1] - c0n;
        s0n = a.tileSlotStart[t]; nsn = a.tileSlotStart[t + 1] - s0n;
        h0n = a.tileHaloStart[t]; nhn = a.tileHaloStart[t + 1] - h0n;
        sl0n = a.tileSliceStart[t]; nsln = a.tileSliceStart[t + 1] - sl0n;
        const double2* src = reinterpret_cast<const double2*>(a.up + s0n);
        const int ns2 = nsn >> 1;
        int hi[HV];
#pragma unroll
        for (int u = 0; u < HV; ++u) { const int k = tid + u * BS; hi[u] = k < nhn ? a.haloCell[h0n + k] : 0; }
#pragma unroll
        for (int u = 0; u < CU4; ++u) { const int k = tid + u * BS; rc[u] = k < ns2 ? src[k] : make_double2(0.0, 0.0); }
#pragma unroll
        for (int u = 0; u < XV; ++u) { const int k = tid + u * BS; rx[u] = k < ncn ? a.x[c0n + k] : 0.0; }
#pragma unroll
        for (int u = 0; u < HV; ++u) { const int k = tid + u * BS; rh[u] = k < nhn ? a.x[hi[u]] : 0.0; }
    };
    prefetch(p0);
    for (int p = p0; p < p1; ++p) {
        // ---- commit the prefetched tile to LDS
        const int c0 = c0n, nc = ncn, ns = nsn, nh = nhn, sl0 = sl0n, nsl = nsln, h0 = h0n;
        __syncthreads(); // everybody is done with the previous image
        {
            double2* dC = reinterpret_cast<double2*>(cU);
            const int ns2 = ns >> 1;
#pragma unroll
            for (int u = 0; u < CU4; ++u) { const int k = tid + u * BS; if (k < ns2) dC[k] = rc[u]; }
#pragma unroll
            for (int u = 0; u < XV; ++u) { const int k = tid + u * BS; if (k < nc) xs[k] = rx[u]; }
#pragma unroll
            for (int u = 0; u < HV; ++u) { const int k = tid + u * BS; if (k < nh) xs[nc + k] = rh[u]; }
            for (int k = tid + HV * BS; k < nh; k += BS) xs[nc + k] = a.x[a.haloCell[h0 + k]]; // rare: very large halos
        }
        // first slice entries of this tile for this wave
        const uint32_t padEnt = (uint32_t)(ns - 1) << 16;
        uint32_t ecur[PRE];
        int wcur = 0, e0cur = 0;
        auto fetch = [&](int s, uint32_t (&e)[PRE], int& e0, int& width) {
            e0 = __builtin_amdgcn_readfirstlane(a.sliceEntryStart[sl0 + s]);
            const int e1 = __builtin_amdgcn_readfirstlane(a.sliceEntryStart[sl0 + s + 1]);
            width = (e1 - e0) >> 6;
            const uint32_t* ent = a.entries + e0 + lane;
#pragma unroll
            for (int q = 0; q < PRE; ++q) e[q] = (q < width) ? ent[q * 64] : padEnt;
        };
        if (wave < nsl) fetch(wave, ecur, e0cur, wcur);
        __syncthreads();
        // ---- the next tile's loads go out now and stay in flight during the row loop
        if (p + 1 < p1) prefetch(p + 1);
        for (int s = wave; s < nsl; s += NW) {
            uint32_t enext[PRE];
            int wnext = 0, e0next = 0;
            if (s + NW < nsl) fetch(s + NW, enext, e0next, wnext);
            else {
#pragma unroll
                for (int q = 0; q < PRE; ++q) enext[q] = padEnt;
            }
            const int i = s * 64 + lane;
            const bool live = i < nc;
            const int gi = c0 + (live ? i : 0);
            const double xi = live ? xs[i] : 0.0;
            double acc = a.diag[gi] * xi;
            auto accumulate = [&](uint32_t en) { acc = fma(cU[(en >> 16) & 0x7FFFu], xs[en & 0xFFFFu], acc); };
#pragma unroll
            for (int q = 0; q < PRE; ++q) if (q < wcur) accumulate(ecur[q]);
            if (wcur > PRE) {
                const uint32_t* ent = a.entries + e0cur + lane;
                for (int q = PRE; q < wcur; ++q) accumulate(ent[q * 64]);
            }
            if (live) a.y[gi] = acc;
#pragma unroll
            for (int q = 0; q < PRE; ++q) ecur[q] = enext[q];
            wcur = wnext; e0cur = e0next;
        }
    }
}

template <class Run, class Check>
void exp_prefetch(TileArgs& a, const TileLayout& L, size_t ldsX, Run run, Check check, double* y1)
{
    static bool once = false;
    if (!once) {
        once = true;
        (void)hipFuncSetAttribute((const void*)k_amul_prefetch<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        (void)hipFuncSetAttribute((const void*)k_amul_prefetch<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_amul_prefetch<512>, 512, ldsX);
        printf("   prefetch kernel BS512: %d workgroups per CU by the runtime\n", nb);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_amul_prefetch<1024>, 1024, ldsX);
        printf("   prefetch kernel BS1024: %d workgroups per CU by the runtime\n", nb);
    }
    for (int wg : {1, 2, 3, 4}) {
        char name[64]; snprintf(name, sizeof(name), "prefetch BS512 %d WG/CU", wg);
        run(name, [&] { k_amul_prefetch<512><<<256 * wg, 512, ldsX, 0>>>(a); }, y1);
        check(name);
    }
    for (int wg : {1, 2}) {
        char name[64]; snprintf(name, sizeof(name), "prefetch BS1024 %d WG/CU", wg);
        run(name, [&] { k_amul_prefetch<1024><<<256 * wg, 1024, ldsX, 0>>>(a); }, y1);
        check(name);
    }
}
} // namespace mi

// ---------------------------------------------------------------------------------------------------------------------
// V4: coefficient segment and the tile's own psi range staged with direct-to-LDS loads (global_load_lds_dwordx4: no VGPR
//     round trip, one instruction per 1 KiB per wave); the halo gather stays on the register path.  One workgroup per tile.
// ---------------------------------------------------------------------------------------------------------------------
namespace mi {
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int BS>
__global__ __launch_bounds__(BS) void k_amul_dma(const TileArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* cU = smem;
    double* xs = smem + a.offX;
    const int b = blockIdx.x, perx = gridDim.x >> 3;
    const int t = (b < (perx << 3)) ? (b & 7) * perx + (b >> 3) : b;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NW = BS / 64, PRE = 8;
    const int c0 = a.tileCellStart[t], nc = a.tileCellStart[t + 1] - c0;
    const int s0 = a.tileSlotStart[t], ns = a.tileSlotStart[t + 1] - s0;
    const int h0 = a.tileHaloStart[t], nh = a.tileHaloStart[t + 1] - h0;
    {   // coefficients: ns/2 double2, wave w of pass q copies elements [q*BS + w*64, +64)
        const double2* src = reinterpret_cast<const double2*>(a.up + s0);
        double2* dst = reinterpret_cast<double2*>(cU);
        const int n2 = ns >> 1;
        for (int base = wave * 64; base < n2; base += BS)
            if (base + lane < n2) __builtin_amdgcn_global_load_lds((gptr_t)(src + base + lane), (lptr_t)(dst + base), 16, 0, 0);
        // psi of the tile's own cells: pairs, the odd last one by hand
        const double2* xsrc = reinterpret_cast<const double2*>(a.x + c0);
        double2* xdst = reinterpret_cast<double2*>(xs);
        const int x2 = nc >> 1;
        for (int base = wave * 64; base < x2; base += BS)
            if (base + lane < x2) __builtin_amdgcn_global_load_lds((gptr_t)(xsrc + base + lane), (lptr_t)(xdst + base), 16, 0, 0);
        if ((nc & 1) && tid == 0) xs[nc - 1] = a.x[c0 + nc - 1];
    }
    stage_gather<BS>(a.x, a.haloCell + h0, xs + nc, nh, tid);
    const int sl0 = a.tileSliceStart[t], nsl = a.tileSliceStart[t + 1] - sl0;
    const uint32_t padEnt = (uint32_t)(ns - 1) << 16;
    uint32_t ecur[PRE];
    int wcur = 0, e0cur = 0;
    auto fetch = [&](int s, uint32_t (&e)[PRE], int& e0, int& width) {
        e0 = __builtin_amdgcn_readfirstlane(a.sliceEntryStart[sl0 + s]);
        const int e1 = __builtin_amdgcn_readfirstlane(a.sliceEntryStart[sl0 + s + 1]);
        width = (e1 - e0) >> 6;
        const uint32_t* ent = a.entries + e0 + lane;
#pragma unroll
        for (int q = 0; q < PRE; ++q) e[q] = (q < width) ? ent[q * 64] : padEnt;
    };
    if (wave < nsl) fetch(wave, ecur, e0cur, wcur);
    __syncthreads();
    for (int s = wave; s < nsl; s += NW) {
        uint32_t enext[PRE];
        int wnext = 0, e0next = 0;
        if (s + NW < nsl) fetch(s + NW, enext, e0next, wnext);
        else {
#pragma unroll
            for (int q = 0; q < PRE; ++q) enext[q] = padEnt;
        }
        const int i = s * 64 + lane;
        const bool live = i < nc;
        const int gi = c0 + (live ? i : 0);
        const double xi = live ? xs[i] : 0.0;
        double acc = a.diag[gi] * xi;
        auto accumulate = [&](uint32_t en) { acc = fma(cU[(en >> 16) & 0x7FFFu], xs[en & 0xFFFFu], acc); };
#pragma unroll
        for (int q = 0; q < PRE; ++q) if (q < wcur) accumulate(ecur[q]);
        if (wcur > PRE) {
            const uint32_t* ent = a.entries + e0cur + lane;
            for (int q = PRE; q < wcur; ++q) accumulate(ent[q * 64]);
        }
        if (live) a.y[gi] = acc;
#pragma unroll
        for (int q = 0; q < PRE; ++q) ecur[q] = enext[q];
        wcur = wnext; e0cur = e0next;
    }
}

template <class Run, class Check>
void exp_dma(TileArgs& a, const TileLayout& L, size_t ldsX, Run run, Check check, double* y1)
{
    static bool once = false;
    if (!once) {
        once = true;
        (void)hipFuncSetAttribute((const void*)k_amul_dma<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024);
        int nb = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_amul_dma<512>, 512, ldsX);
        printf("   direct-to-LDS kernel BS512: %d workgroups per CU by the runtime\n", nb);
    }
    run("direct-to-LDS staging BS512", [&] { k_amul_dma<512><<<L.nTiles, 512, ldsX, 0>>>(a); }, y1);
    check("direct-to-LDS");
}
} // namespace mi
