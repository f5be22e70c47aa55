// Host-only measurement (round 4): do the GAMG level layouts lose quality when level l + 1 INHERITS its tiles from level l
// (csrc/tiling.hpp inherit_tiles + TileParams::givenPart) instead of running the multilevel clustering again?  The clustering's
// sequential heavy-edge matching inside the first level's layout is the critical path of mi_gamg_create (DESIGN 3.4 "Round 4 (v)").
// For the levels of the nx^3 box: tiles / halo entries / coefficient slots and the host seconds of
//   A  the layout the engine builds today (clustered from scratch, 1024-cell tiles),
//   B  the layout of the inherited tiles.
//   g++ -O3 -std=c++17 -pthread -Irapidcfd-dev_amd/csrc tools/exp/inherit_tiles.cpp rapidcfd-dev_amd/csrc/tiling.cpp rapidcfd-dev_amd/csrc/gamg.cpp -o tools/exp/inherit_tiles && tools/exp/inherit_tiles 216
#include "gamg.hpp"
#include "tiling.hpp"
#include <chrono>
#include <cstdio>
#include <cstdlib>
using namespace mi;
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void report(const char* tag, const TileLayout& L, double sec)
{
    printf("    %-24s tiles %6d  halo entries %8zu  coefficient slots %9zu  largest halo %4d   %.3f s\n", tag, L.nTiles, L.haloCell.size(), L.slotFace.size(), L.maxHalo, sec);
}
static std::vector<int32_t> tile_of_cell(const TileLayout& L, int32_t n)
{
    std::vector<int32_t> t((size_t)n);
    for (int32_t k = 0; k < L.nTiles; ++k) for (int32_t e = L.tileCellStart[(size_t)k]; e < L.tileCellStart[(size_t)k + 1]; ++e) t[(size_t)L.e2c[(size_t)e]] = k;
    return t;
}
int main(int argc, char** argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 216, ny = nx, nz = nx;
    const int32_t n = nx * ny * nz;
    std::vector<int32_t> lo, up; std::vector<double> w;
    for (int z = 0; z < nz; ++z) for (int y = 0; y < ny; ++y) for (int x = 0; x < nx; ++x) {
        const int32_t c = (z * ny + y) * nx + x;
        if (x + 1 < nx) { lo.push_back(c); up.push_back(c + 1); w.push_back(1.0 / nx); }
        if (y + 1 < ny) { lo.push_back(c); up.push_back(c + nx); w.push_back(1.01 / nx); }
        if (z + 1 < nz) { lo.push_back(c); up.push_back(c + nx * ny); w.push_back(1.02 / nx); }
    }
    TileParams prm; prm.compact = false;
    TileLayout L0;
    double t0 = now();
    std::string err = build_tile_layout(n, (int32_t)lo.size(), lo.data(), up.data(), 0, nullptr, nullptr, prm, L0, nullptr);
    if (!err.empty()) { printf("%s\n", err.c_str()); return 1; }
    printf("finest level: %d cells\n", n); report("clustered", L0, now() - t0);
    GamgHierarchyHost H;
    err = build_gamg_hierarchy(n, (int32_t)lo.size(), lo.data(), up.data(), w.data(), 100, true, H, nullptr, 1, 0);
    if (!err.empty()) { printf("%s\n", err.c_str()); return 1; }
    std::vector<int32_t> fineTile = tile_of_cell(L0, n);
    int32_t nFineTiles = L0.nTiles;
    for (size_t l = 0; l < H.levels.size() && H.levels[l].nCoarse > 20000; ++l) {
        const GamgLevelHost& G = H.levels[l];
        printf("level %zu: %d cells, %d faces\n", l + 1, G.nCoarse, G.nCoarseFaces);
        TileLayout A, B;
        t0 = now();
        err = build_tile_layout(G.nCoarse, G.nCoarseFaces, G.cLower.data(), G.cUpper.data(), 0, nullptr, nullptr, prm, A, nullptr);
        if (!err.empty()) { printf("%s\n", err.c_str()); return 1; }
        report("A clustered (today)", A, now() - t0);
        t0 = now();
        std::vector<int32_t> part; int32_t nParts = 0;
        err = inherit_tiles(G.nFine, G.restrictMap.data(), fineTile.data(), nFineTiles, G.nCoarse, G.nCoarseFaces, G.cLower.data(), G.cUpper.data(), 0, nullptr, nullptr,
                            prm.tileCells, prm.slotCap, part, nParts);
        if (!err.empty()) { printf("    B: %s\n", err.c_str()); return 1; }
        TileParams pg = prm; pg.givenPart = part.data(); pg.nGivenParts = nParts;
        err = build_tile_layout(G.nCoarse, G.nCoarseFaces, G.cLower.data(), G.cUpper.data(), 0, nullptr, nullptr, pg, B, nullptr);
        if (!err.empty()) { printf("    B: %s\n", err.c_str()); return 1; }
        report("B inherited tiles", B, now() - t0);
        const bool same = A.e2c == B.e2c && A.tileCellStart == B.tileCellStart && A.entries == B.entries && A.slotFace == B.slotFace;
        printf("    %s\n", same ? "B is table for table the layout A" : "(different tiles)");
        fineTile.swap(part); nFineTiles = nParts;
    }
    return 0;
}
