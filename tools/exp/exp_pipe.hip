// Experiment harness (not a product path): PIPELINED persistent Amul tile kernel.
//   * every workgroup walks a contiguous run of tiles of its XCD;
//   * two LDS stages per workgroup: while the rows of tile k are computed out of stage k&1, the image of tile k+1
//     (coefficients, own psi range, halo) is in flight into the other stage as LDS-DMA (global_load_lds_dwordx4 issued
//     from inline asm, so hipcc's waitcnt bookkeeping does not drain it in front of the ds_reads of tile k);
//   * the halo is gathered by DMA too: the halo list holds PAIRS of adjacent engine cells, one 16-byte per-lane fetch
//     each; the entries' `other` field addresses the pair image (ncE + 2q + off);
//   * row entries and the diagonal of tile k+1 are prefetched into registers during tile k (compiler-visible loads that
//     are complete at the top-of-loop vmcnt(0), so the compute phase never waits on vector memory).
// Build: make -C tools/exp exp_pipe ; run: TILE_CELLS=512 tools/exp/exp_pipe [n] [reps]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../rapidcfd-dev_amd/csrc/kernels.hip.hpp"
#include "../../rapidcfd-dev_amd/csrc/tiling.hpp"
using namespace mi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
template <class T> T* up(const std::vector<T>& v) { T* p = nullptr; CK(hipMalloc(&p, sizeof(T) * (v.size() + 64))); CK(hipMemset(p, 0, sizeof(T) * (v.size() + 64))); CK(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice)); return p; }

struct PipeArgs {
    const int32_t* desc;            // [nPos][8]: c0, nc, s0, ns, q0, nq, sl0, nsl
    const int32_t* sliceEntryStart;
    const uint32_t* entries;        // other (pair-image addressing) | slot<<16 | isLower<<31
    const int32_t* pairStart;       // engine index of the first cell of every halo pair
    const double *diag, *up, *x;
    double* y;
    double* dotPartial;
    int32_t nPos;
    int32_t stageD;                 // doubles per LDS stage
    int32_t offX;                   // doubles: own psi range inside a stage (coefficients at 0)
    int32_t idxOffD;                // doubles from smem to the pair-index ring
    int32_t idxCap;                 // int32 per ring slot (multiple of 4)
    int32_t redOffD;                // doubles from smem to the per-wave dot partials [2][NW]
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(lds_ptr_t)p; }

template <bool NT>
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t ldsDst)
{
    uint32_t keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
}
__device__ __forceinline__ void dma4(const void* gsrc, uint32_t ldsDst)
{
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(ldsDst) : "memory");
}

// contiguous copy of n16 16-byte units; DMA op j (64 units) is issued by wave (opBase + j) % NW
template <int NW, bool NT>
__device__ __forceinline__ int copy_region(const char* src, uint32_t ldsBase, int n16, int opBase, int wave, int lane)
{
    const int nOps = (n16 + 63) >> 6;
    int j = wave - (opBase % NW); if (j < 0) j += NW;
    for (; j < nOps; j += NW) {
        const int u = (j << 6) + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(ldsBase + (uint32_t)(j << 10));
        if (u < n16) dma16<NT>(src + ((size_t)u << 4), dst);
    }
    return opBase + nOps;
}

// layout tables are immutable while a kernel runs: reading them through the constant address space lets hipcc use scalar
// loads (s_load, counted by lgkmcnt) although the kernel also stores to global memory and issues asm with a memory clobber
typedef __attribute__((address_space(4))) const int32_t* cint_ptr_t;
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(4))) const v4i_t* cint4_ptr_t;
__device__ __forceinline__ int32_t cload(const int32_t* p, int i) { return ((cint_ptr_t)(uintptr_t)p)[i]; }
struct TDesc { int32_t c0, nc, s0, ns, q0, nq, sl0, nsl; };
__device__ __forceinline__ TDesc load_desc(const int32_t* desc, int p)
{
    const cint4_ptr_t q = (cint4_ptr_t)(uintptr_t)(desc + 8 * p);
    const v4i_t lo = q[0], hi = q[1];
    TDesc d; d.c0 = lo.x; d.nc = lo.y; d.s0 = lo.z; d.ns = lo.w; d.q0 = hi.x; d.nq = hi.y; d.sl0 = hi.z; d.nsl = hi.w;
    return d;
}

template <int BS, int SPW, bool NT>
__global__ __launch_bounds__(BS) void k_amul_pipe(const PipeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NW = BS / 64, PRE = 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x, G = gridDim.x, nT = a.nPos;
    const int per = G >> 3, xc = b & 7, jb = b >> 3;
    const int x0 = (int)((long long)xc * nT / 8), x1 = (int)((long long)(xc + 1) * nT / 8);
    const int p0 = x0 + (int)((long long)jb * (x1 - x0) / per), p1 = x0 + (int)((long long)(jb + 1) * (x1 - x0) / per);
    if (p0 >= p1) return;
    int32_t* idxRing = reinterpret_cast<int32_t*>(smem + a.idxOffD);
    double* red = smem + a.redOffD;
    const uint32_t smemA = lds_addr(smem), idxA = lds_addr(idxRing);

    auto issue_idx = [&](const TDesc& d, int slot) { // pair indices of a tile -> ring slot
        copy_region<NW, false>(reinterpret_cast<const char*>(a.pairStart + d.q0), idxA + (uint32_t)(slot * a.idxCap * 4), (d.nq + 3) >> 2, 0, wave, lane);
    };
    auto issue_image = [&](const TDesc& d, int stage, int slot) { // coefficients, own psi, halo pairs of a tile -> LDS stage
        const uint32_t stA = smemA + (uint32_t)(stage * a.stageD * 8);
        int ob = copy_region<NW, NT>(reinterpret_cast<const char*>(a.up + d.s0), stA, d.ns >> 1, 0, wave, lane);
        ob = copy_region<NW, false>(reinterpret_cast<const char*>(a.x + d.c0), stA + (uint32_t)(a.offX * 8), d.nc >> 1, ob, wave, lane);
        if ((d.nc & 1) && wave == (ob % NW)) { // odd tail: the last double as two dwords
            const uint32_t dst = __builtin_amdgcn_readfirstlane(stA + (uint32_t)((a.offX + d.nc - 1) * 8));
            if (lane < 2) dma4(reinterpret_cast<const char*>(a.x + d.c0 + d.nc - 1) + 4 * lane, dst);
        }
        ob += 1;
        const int ncE = (d.nc + 1) & ~1;
        const int32_t* ring = idxRing + slot * a.idxCap;
        const int nOps = (d.nq + 63) >> 6;
        int j = wave - (ob % NW); if (j < 0) j += NW;
        for (; j < nOps; j += NW) {
            const int q = (j << 6) + lane;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(stA + (uint32_t)((a.offX + ncE) * 8) + (uint32_t)(j << 10));
            if (q < d.nq) dma16<false>(a.x + ring[q], dst);
        }
    };

    // registers that carry the row entries / diagonal of the NEXT tile, and the entry offsets of the one after
    uint32_t ecur[SPW][PRE], enext[SPW][PRE];
    double dcur[SPW], dnext[SPW];
    int wcur[SPW], wnext[SPW], e0cur[SPW], e0next[SPW];
    int sesA[SPW][2], sesB[SPW][2];
    auto load_ses = [&](const TDesc& d, int (&ses)[SPW][2]) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int sl = wave + s * NW;
            ses[s][0] = 0; ses[s][1] = 0;
            if (sl < d.nsl) { ses[s][0] = cload(a.sliceEntryStart, d.sl0 + sl); ses[s][1] = cload(a.sliceEntryStart, d.sl0 + sl + 1); }
        }
    };
    auto prefetch_rows = [&](const TDesc& d, const int (&ses)[SPW][2], uint32_t (&e)[SPW][PRE], double (&dg)[SPW], int (&w)[SPW], int (&e0)[SPW]) {
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int sl = wave + s * NW;
            e0[s] = ses[s][0]; w[s] = (ses[s][1] - ses[s][0]) >> 6; dg[s] = 0.0;
            const uint32_t* ent = a.entries + e0[s] + lane;
            const uint32_t padEnt = (uint32_t)(d.ns - 1) << 16; // the last slot of a segment is always 0.0
#pragma unroll
            for (int j = 0; j < PRE; ++j) e[s][j] = (j < w[s]) ? ent[j * 64] : padEnt;
            const int i = sl * 64 + lane;
            if (i < d.nc) dg[s] = a.diag[d.c0 + i];
        }
    };

    // ---- prologue ---------------------------------------------------------------------------------------------
    TDesc d0 = load_desc(a.desc, p0), dA = d0, dB = d0;
    if (p0 + 1 < p1) dA = load_desc(a.desc, p0 + 1);
    issue_idx(d0, 0);
    if (p0 + 1 < p1) issue_idx(dA, 1);
    load_ses(d0, sesB);
    load_ses(dA, sesA);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    issue_image(d0, 0, 0);
    prefetch_rows(d0, sesB, ecur, dcur, wcur, e0cur);

    for (int p = p0; p < p1; ++p) {
        const int k = p - p0, st = k & 1;
        if (p + 2 < p1) dB = load_desc(a.desc, p + 2);
        __builtin_amdgcn_s_waitcnt(0x0F70);           // vmcnt(0): image p (+ pair indices p+1, row registers p) have landed
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // ... for every wave; and every wave is done with tile p-1
        if (k > 0 && a.dotPartial && tid == 0) {       // per-tile partial of tile p-1 from its wave sums, in wave order
            const double* r = red + (st ^ 1) * NW;
            double t = r[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += r[w];
            a.dotPartial[p - 1] = t;
        }
        if (p + 1 < p1) {
            issue_image(dA, st ^ 1, st ^ 1);
            if (p + 2 < p1) issue_idx(dB, st);
            prefetch_rows(dA, sesA, enext, dnext, wnext, e0next);
            if (p + 2 < p1) load_ses(dB, sesB);
        }
        // ---- rows of tile p out of stage st: LDS only ------------------------------------------------------------
        const double* cU = smem + st * a.stageD;
        const double* xs = cU + a.offX;
        double dot = 0.0;
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int sl = wave + s * NW;
            if (sl < d0.nsl) {
                const int i = sl * 64 + lane;
                const bool live = i < d0.nc;
                const double xi = live ? xs[i] : 0.0;
                double acc = dcur[s] * xi;
                // groups of four entries, all LDS reads of a group in flight together (padding entries read the zero slot)
#pragma unroll
                for (int g = 0; g < PRE; g += 4)
                    if (g < wcur[s]) {
                        double cc[4], xx[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) { const uint32_t en = ecur[s][g + j]; cc[j] = cU[(en >> 16) & 0x7FFFu]; xx[j] = xs[en & 0xFFFFu]; }
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc = fma(cc[j], xx[j], acc);
                    }
                if (wcur[s] > PRE) {
                    const uint32_t* ent = a.entries + e0cur[s] + lane;
                    for (int j = PRE; j < wcur[s]; ++j) { const uint32_t en = ent[j * 64]; acc = fma(cU[(en >> 16) & 0x7FFFu], xs[en & 0xFFFFu], acc); }
                }
                if (live) { a.y[d0.c0 + i] = acc; dot = fma(acc, xi, dot); }
            }
        }
        if (a.dotPartial) { const double ws = wave_sum(dot); if (lane == 0) red[st * NW + wave] = ws; }
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
#pragma unroll
            for (int j = 0; j < PRE; ++j) ecur[s][j] = enext[s][j];
            dcur[s] = dnext[s]; wcur[s] = wnext[s]; e0cur[s] = e0next[s];
            sesA[s][0] = sesB[s][0]; sesA[s][1] = sesB[s][1];
        }
        d0 = dA; dA = dB;
    }
    if (a.dotPartial) {
        __syncthreads();
        if (tid == 0) {
            const double* r = red + ((p1 - 1 - p0) & 1) * NW;
            double t = r[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) t += r[w];
            a.dotPartial[p1 - 1] = t;
        }
    }
}

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 216;
    const int reps = argc > 2 ? atoi(argv[2]) : 40;
    const int N = n * n * n;
    std::vector<int32_t> lower, upper;
    for (int k = 0; k < n; ++k) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        const int c = (k * n + j) * n + i;
        if (i + 1 < n) { lower.push_back(c); upper.push_back(c + 1); }
        if (j + 1 < n) { lower.push_back(c); upper.push_back(c + n); }
        if (k + 1 < n) { lower.push_back(c); upper.push_back(c + n * n); }
    }
    const int F = (int)lower.size();
    TileLayout L; TileParams prm; prm.compact = false;
    if (getenv("TILE_CELLS")) prm.tileCells = atoi(getenv("TILE_CELLS"));
    if (getenv("TILE_SLOTS")) prm.slotCap = atoi(getenv("TILE_SLOTS"));
    const std::string err = build_tile_layout(N, F, lower.data(), upper.data(), 0, nullptr, nullptr, prm, L);
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    printf("N=%d F=%d tiles=%d slots=%lld maxCells=%d maxSlots=%d maxHalo=%d\n", N, F, L.nTiles, (long long)L.totalSlots, L.maxCells, L.maxSlots, L.maxHalo);

    // ---- pair halo lists + remapped entries (host post-processing of the current layout) ---------------------------
    const int nT = L.nTiles;
    std::vector<int32_t> pairStart, desc((size_t)nT * 8);
    std::vector<uint32_t> ent2 = L.entries;
    int maxPairs = 0; long long totPairs = 0;
    for (int t = 0; t < nT; ++t) {
        const int c0 = L.tileCellStart[t], nc = L.tileCellStart[t + 1] - c0;
        const int h0 = L.tileHaloStart[t], nh = L.tileHaloStart[t + 1] - h0;
        std::vector<std::pair<int32_t, int32_t>> cells((size_t)nh); // (engine cell, halo index)
        for (int h = 0; h < nh; ++h) cells[(size_t)h] = {L.haloCell[(size_t)h0 + h], h};
        std::sort(cells.begin(), cells.end());
        std::vector<int32_t> hmap((size_t)nh, -1);
        const int q0 = (int)pairStart.size();
        for (size_t k = 0; k < cells.size();) {
            const int32_t c = cells[k].first;
            const int q = (int)pairStart.size() - q0;
            if (k + 1 < cells.size() && cells[k + 1].first == c + 1) { pairStart.push_back(c); hmap[(size_t)cells[k].second] = 2 * q; hmap[(size_t)cells[k + 1].second] = 2 * q + 1; k += 2; }
            else if (c + 1 < N) { pairStart.push_back(c); hmap[(size_t)cells[k].second] = 2 * q; k += 1; }
            else { pairStart.push_back(c - 1); hmap[(size_t)cells[k].second] = 2 * q + 1; k += 1; }
        }
        const int nq = (int)pairStart.size() - q0;
        while (pairStart.size() & 3u) pairStart.push_back(0); // ring slots are filled in 16-byte units
        maxPairs = std::max(maxPairs, nq); totPairs += nq;
        const int ncE = (nc + 1) & ~1;
        const int sl0 = L.tileSliceStart[t], nsl = L.tileSliceStart[t + 1] - sl0;
        for (int e = L.sliceEntryStart[sl0]; e < L.sliceEntryStart[sl0 + nsl]; ++e) {
            const uint32_t en = ent2[(size_t)e]; const int o = (int)(en & 0xFFFFu);
            if (o >= nc) ent2[(size_t)e] = (en & 0xFFFF0000u) | (uint32_t)(ncE + hmap[(size_t)(o - nc)]);
        }
        int32_t* d = &desc[(size_t)t * 8];
        d[0] = c0; d[1] = nc; d[2] = L.tileSlotStart[t]; d[3] = L.tileSlotStart[t + 1] - L.tileSlotStart[t]; d[4] = q0; d[5] = nq; d[6] = sl0; d[7] = nsl;
    }
    printf("halo cells %zu -> pairs %lld (max %d per tile); entries %zu words\n", L.haloCell.size(), totPairs, maxPairs, L.entries.size());

    std::vector<double> coef((size_t)L.totalSlots, 0.0), diag((size_t)N), x((size_t)N);
    for (size_t s = 0; s < coef.size(); ++s) if (L.slotFace[s] >= 0) coef[s] = -1.0 - 1e-3 * (L.slotFace[s] % 97);
    for (int c = 0; c < N; ++c) { diag[c] = 6.5 + 1e-3 * (c % 13); x[c] = 0.5 + 1e-4 * (c % 1001); }

    // baseline: the engine's tile kernel on the unchanged layout
    TileArgs a; memset(&a, 0, sizeof(a));
    a.tileCellStart = up(L.tileCellStart); a.tileSlotStart = up(L.tileSlotStart); a.tileIfaceSlot0 = up(L.tileIfaceSlot0);
    a.tileHaloStart = up(L.tileHaloStart); a.haloCell = up(L.haloCell); a.tileSliceStart = up(L.tileSliceStart);
    a.sliceEntryStart = up(L.sliceEntryStart); a.entries = up(L.entries);
    a.nPos = L.nTiles; a.diag = up(diag); a.up = up(coef); a.low = a.up; a.x = up(x);
    double *y0, *y1, *dp0, *dp1;
    CK(hipMalloc(&y0, sizeof(double) * N)); CK(hipMalloc(&y1, sizeof(double) * N));
    CK(hipMalloc(&dp0, sizeof(double) * nT)); CK(hipMalloc(&dp1, sizeof(double) * nT));
    const int slots = (L.maxSlots + 3) & ~1, xlen = ((L.maxCells + 63) & ~63) + L.maxHalo + 2;
    a.offLow = slots; a.offX = slots; a.offRD = slots + ((xlen + 1) & ~1); a.offSB = a.offRD;
    const size_t ldsX = (size_t)a.offRD * 8;
    a.dotPartial = dp0;

    PipeArgs pa; memset(&pa, 0, sizeof(pa));
    pa.desc = up(desc); pa.sliceEntryStart = a.sliceEntryStart; pa.entries = up(ent2); pa.pairStart = up(pairStart);
    pa.diag = a.diag; pa.up = a.up; pa.x = a.x; pa.y = y1; pa.dotPartial = dp1; pa.nPos = nT;
    pa.offX = slots;
    const int xlenP = ((L.maxCells + 1) & ~1) + 2 * ((maxPairs + 63) & ~63) + 2;
    pa.stageD = (slots + xlenP + 1) & ~1;
    pa.idxOffD = 2 * pa.stageD;
    pa.idxCap = (maxPairs + 67) & ~3;
    pa.redOffD = pa.idxOffD + (2 * pa.idxCap + 1) / 2 + 1;
    const size_t ldsP = (size_t)(pa.redOffD + 2 * 16) * 8;
    printf("LDS: baseline image %zu B; pipelined workgroup %zu B (stage %d B)\n", ldsX, ldsP, pa.stageD * 8);

    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double algBytes = 24.0 * N + 16.0 * F;
    auto run = [&](const char* name, auto launch) {
        for (int r = 0; r < 3; ++r) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-44s %8.1f us   %6.0f GB/s algorithmic  (%.1f %% of 8 TB/s)\n", name, us, algBytes / us * 1e-3, algBytes / us * 1e-3 / 80.0);
        fflush(stdout);
    };
    auto check = [&](const char* name) {
        std::vector<double> h0((size_t)N), h1((size_t)N), d0((size_t)nT), d1((size_t)nT);
        CK(hipMemcpy(h0.data(), y0, sizeof(double) * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, sizeof(double) * N, hipMemcpyDeviceToHost));
        CK(hipMemcpy(d0.data(), dp0, sizeof(double) * nT, hipMemcpyDeviceToHost)); CK(hipMemcpy(d1.data(), dp1, sizeof(double) * nT, hipMemcpyDeviceToHost));
        size_t bad = 0, badd = 0; for (int c = 0; c < N; ++c) if (memcmp(&h0[c], &h1[c], 8) != 0) ++bad;
        double s0 = 0, s1 = 0; for (int t = 0; t < nT; ++t) { s0 += d0[t]; s1 += d1[t]; if (d0[t] != d1[t]) ++badd; }
        printf("   %s vs baseline: %zu of %d values differ; tile partials differing %zu of %d (sum %.17g vs %.17g)\n", name, bad, N, badd, nT, s0, s1);
        CK(hipMemset(y1, 0, sizeof(double) * N)); CK(hipMemset(dp1, 0, sizeof(double) * nT));
    };
    CK(hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
#define SETA(BS, SPW, NT) CK(hipFuncSetAttribute((const void*)k_amul_pipe<BS, SPW, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    SETA(256, 1, false); SETA(256, 2, false); SETA(256, 4, false); SETA(512, 1, false); SETA(512, 2, false); SETA(1024, 1, false);
    SETA(256, 1, true); SETA(256, 2, true); SETA(256, 4, true); SETA(512, 1, true); SETA(512, 2, true); SETA(1024, 1, true);
    const int nsl = (L.maxCells + 63) / 64;
    const int wgMax = (int)((160 * 1024) / ldsP);
    printf("slices per tile %d; pipelined workgroups per CU by LDS: %d\n", nsl, wgMax);
    a.y = y0;
    for (int pass = 0; pass < 2; ++pass) {
        run("baseline tile_kernel BS512", [&] { tile_kernel<OP_AMUL, false, false, 512, false><<<nT, 512, ldsX, 0>>>(a); });
        run("baseline tile_kernel BS256", [&] { tile_kernel<OP_AMUL, false, false, 256, false><<<nT, 256, ldsX, 0>>>(a); });
#define RUNP(BS, SPW, NT, WG)                                                                                             \
        if (nsl <= SPW * (BS / 64) && WG <= wgMax && (WG) * (BS / 64) <= 32) {                                              \
            char nm[96]; snprintf(nm, sizeof nm, "pipe BS%d SPW%d %s wg/CU %d", BS, SPW, NT ? "nt" : "  ", WG);                \
            const int grid = 256 * WG;                                                                                   \
            run(nm, [&] { k_amul_pipe<BS, SPW, NT><<<grid, BS, ldsP, 0>>>(pa); });                                        \
            if (pass == 0) check(nm);                                                                                    \
        }
        for (int wg = 1; wg <= 8; ++wg) {
            RUNP(256, 1, false, wg) RUNP(256, 2, false, wg) RUNP(256, 4, false, wg)
            RUNP(512, 1, false, wg) RUNP(512, 2, false, wg) RUNP(1024, 1, false, wg)
            RUNP(256, 1, true, wg) RUNP(256, 2, true, wg) RUNP(256, 4, true, wg)
            RUNP(512, 1, true, wg) RUNP(512, 2, true, wg) RUNP(1024, 1, true, wg)
        }
    }
    return 0;
}
