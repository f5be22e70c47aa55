// Experiment harness (not a product path): the Amul tile kernel on a synthetic box, standalone, so kernel variants
// can be compiled in seconds and timed on one box back to back.  Build:  make -C tools/exp ; run: tools/exp/exp_tile [n]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../../rapidcfd-dev_amd/csrc/kernels.hip.hpp"
#include "../../rapidcfd-dev_amd/csrc/tiling.hpp"
#include "exp_variants.hpp"
using namespace mi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
template <class T> T* up(const std::vector<T>& v) { T* p = nullptr; CK(hipMalloc(&p, sizeof(T) * (v.size() + 8))); CK(hipMemcpy(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice)); return p; }

int main(int argc, char** argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 216;
    const int reps = argc > 2 ? atoi(argv[2]) : 50;
    const int N = n * n * n;
    std::vector<int32_t> lower, upper;
    for (int k = 0; k < n; ++k) for (int j = 0; j < n; ++j) for (int i = 0; i < n; ++i) {
        const int c = (k * n + j) * n + i;
        if (i + 1 < n) { lower.push_back(c); upper.push_back(c + 1); }
        if (j + 1 < n) { lower.push_back(c); upper.push_back(c + n); }
        if (k + 1 < n) { lower.push_back(c); upper.push_back(c + n * n); }
    }
    const int F = (int)lower.size();
    TileLayout L; TileParams prm;
    if (getenv("TILE_CELLS")) prm.tileCells = atoi(getenv("TILE_CELLS"));
    if (getenv("TILE_SLOTS")) prm.slotCap = atoi(getenv("TILE_SLOTS"));
    const std::string err = build_tile_layout(N, F, lower.data(), upper.data(), 0, nullptr, nullptr, prm, L);
    if (!err.empty()) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    printf("N=%d F=%d tiles=%d slots=%lld maxSlots=%d maxHalo=%d compact=%d\n", N, F, L.nTiles, (long long)L.totalSlots, L.maxSlots, L.maxHalo, (int)L.compact);
    std::vector<double> coef((size_t)L.totalSlots, 0.0), diag((size_t)N), x((size_t)N);
    for (size_t s = 0; s < coef.size(); ++s) if (L.slotFace[s] >= 0) coef[s] = -1.0 - 1e-3 * (L.slotFace[s] % 97);
    for (int c = 0; c < N; ++c) { diag[c] = 6.5 + 1e-3 * (c % 13); x[c] = 0.5 + 1e-4 * (c % 1001); }
    TileArgs a; memset(&a, 0, sizeof(a));
    a.tileCellStart = up(L.tileCellStart); a.tileSlotStart = up(L.tileSlotStart); a.tileIfaceSlot0 = up(L.tileIfaceSlot0);
    a.tileHaloStart = up(L.tileHaloStart); a.haloCell = up(L.haloCell); a.tileSliceStart = up(L.tileSliceStart);
    a.sliceEntryStart = up(L.sliceEntryStart); a.entries = up(L.entries);
    a.entries16 = up(L.entries16); a.sliceEntryStart16 = up(L.sliceEntryStart16);
    a.slotBase = reinterpret_cast<const uint32_t*>(up(L.slotBase)); a.tileSbStart = up(L.tileSbStart);
    a.nPos = L.nTiles; a.diag = up(diag); a.up = up(coef); a.low = a.up; a.x = up(x);
    double *y0, *y1; CK(hipMalloc(&y0, sizeof(double) * N)); CK(hipMalloc(&y1, sizeof(double) * N));
    const int slots = (L.maxSlots + 3) & ~1, xlen = ((L.maxCells + 63) & ~63) + L.maxHalo + 2;
    a.offLow = slots; a.offX = slots; a.offRD = slots + ((xlen + 1) & ~1); a.offSB = a.offRD;
    const size_t lds = (size_t)(a.offSB + (L.maxCells + L.maxHalo + 8) / 4 + 1) * 8;
    const size_t ldsX = (size_t)a.offRD * 8; // explicit entries: no slotBase table
    printf("lds per tile image %zu bytes (explicit entries %zu)\n", lds, ldsX);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const double algBytes = 24.0 * N + 16.0 * F;
    auto run = [&](const char* name, auto launch, double* y) {
        a.y = y;
        for (int r = 0; r < 3; ++r) launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipGetLastError());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        printf("%-28s %8.1f us   %6.0f GB/s algorithmic  (%.1f %% of 8 TB/s)\n", name, us, algBytes / us * 1e-3, algBytes / us * 1e-3 / 80.0);
    };
    auto check = [&](const char* name) {
        std::vector<double> h0((size_t)N), h1((size_t)N);
        CK(hipMemcpy(h0.data(), y0, sizeof(double) * N, hipMemcpyDeviceToHost)); CK(hipMemcpy(h1.data(), y1, sizeof(double) * N, hipMemcpyDeviceToHost));
        size_t bad = 0; for (int c = 0; c < N; ++c) if (memcmp(&h0[c], &h1[c], 8) != 0) ++bad;
        printf("   %s vs baseline: %zu of %d values differ\n", name, bad, N);
    };
    CK(hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 1024, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    CK(hipFuncSetAttribute((const void*)tile_kernel<OP_AMUL, false, false, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    if (getenv("EXP_SWEEP")) { // entry form x workgroup size x cache-policy flags (bit0 nt coefficients, bit1 nt entries, bit2 nt stores)
        a.flags = 0;
        run("reference: explicit BS512 flags 0", [&] { tile_kernel<OP_AMUL, false, false, 512, false><<<L.nTiles, 512, ldsX, 0>>>(a); }, y0);
        for (int pass = 0; pass < 2; ++pass)
            for (int form = 0; form < (L.compact ? 2 : 1); ++form)
                for (int bs : {256, 512})
                    for (int fl : {0, 1, 2, 3, 7}) {
                        a.flags = fl;
                        char nm[96]; snprintf(nm, sizeof nm, "%s BS%d flags %d", form ? "compact " : "explicit", bs, fl);
                        const size_t l = form ? lds : ldsX;
                        if (form == 0 && bs == 256) run(nm, [&] { tile_kernel<OP_AMUL, false, false, 256, false><<<L.nTiles, 256, l, 0>>>(a); }, y1);
                        if (form == 0 && bs == 512) run(nm, [&] { tile_kernel<OP_AMUL, false, false, 512, false><<<L.nTiles, 512, l, 0>>>(a); }, y1);
                        if (form == 1 && bs == 256) run(nm, [&] { tile_kernel<OP_AMUL, false, false, 256, true><<<L.nTiles, 256, l, 0>>>(a); }, y1);
                        if (form == 1 && bs == 512) run(nm, [&] { tile_kernel<OP_AMUL, false, false, 512, true><<<L.nTiles, 512, l, 0>>>(a); }, y1);
                        if (pass == 0) check(nm);
                    }
        return 0;
    }
    for (int pass = 0; pass < 2; ++pass) {
        run("explicit BS512 (own LDS)", [&] { tile_kernel<OP_AMUL, false, false, 512, false><<<L.nTiles, 512, ldsX, 0>>>(a); }, y0);
        run("explicit BS256 (own LDS)", [&] { tile_kernel<OP_AMUL, false, false, 256, false><<<L.nTiles, 256, ldsX, 0>>>(a); }, y1);
        run("explicit BS1024 (own LDS)", [&] { tile_kernel<OP_AMUL, false, false, 1024, false><<<L.nTiles, 1024, ldsX, 0>>>(a); }, y1);
        run("baseline 512 explicit", [&] { tile_kernel<OP_AMUL, false, false, 512, false><<<L.nTiles, 512, lds, 0>>>(a); }, y0);
        exp_variants(a, L, lds, run, check, y1);
        if (getenv("EXP_PREFETCH")) exp_prefetch(a, L, ldsX, run, check, y1);
        exp_dma(a, L, ldsX, run, check, y1);
    }
    return 0;
}
