// tiling.hpp -- host-side construction of the engine's tiled LDU layout.
//
// Replaces the reference's demand-driven lduAddressing tables (losort,
// ownerStart, losortStart, ownerSortAddr, patchSort*; lduAddressing.H:128-145,
// lduAddressing.C:169-400 -- K23 in SURVEY.md) with a layout designed for
// MI355X: cells are clustered into compact tiles (<= TILE_CELLS cells) by
// multilevel heavy-edge matching on the face graph, renumbered tile by tile,
// and every tile gets
//   * a contiguous "slot" segment of face coefficients (each internal face once,
//     cut faces once per side) that one workgroup streams into LDS with wide
//     coalesced loads,
//   * a halo list (cells of other tiles / other ranks it reads),
//   * per-row entry lists {slot, other cell, side} stored in 64-row slices
//     column-major so a wavefront reads them coalesced and loops uniformly
//     (32-bit explicit form, and a 16-bit form with implied slots).
// Entries of a row are ordered exactly like the reference's row gather
// (lduMatrixATmul.C:90-136: upper faces ascending, then lower faces in losort
// order, then the coupled interfaces in patch order) so the floating-point
// summation order is preserved.
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "host_tables.hpp"

namespace mi {

struct TileLayout {
    int32_t nCells = 0, nFaces = 0, nExt = 0, nTiles = 0, nSlices = 0;
    int32_t nPatches = 0;
    Table<int32_t> e2c, c2e;        // engine<->caller cell permutation
    Table<int32_t> tileCellStart;   // [nTiles+1] engine cell range
    Table<int32_t> tileSlotStart;   // [nTiles+1] slot range (starts are even)
    Table<int32_t> tileIfaceSlot0;  // [nTiles] local slot index of the first interface slot (they follow the face slots)
    Table<int32_t> tileHaloStart;   // [nTiles+1]
    Table<int32_t> haloCell;        // engine index; >= nCells means ext cell
    Table<int32_t> tileSliceStart;  // [nTiles+1]
    Table<int32_t> sliceEntryStart; // [nSlices+1]
    Table<uint32_t> entries;        // other | slot<<16 | isLower<<31
    // compact form (half the bytes), usable when every tile has <= 4095 cells+halo and no cell owns more than 8 faces
    // of one tile: 16-bit entries {other:12 | k:3 | rule:1}, two per word.  The slot is implied by the slot ORDER of a
    // tile: rule 0 (the row owns the face; always the leading entries of a row) slot = slotBase[row] + position in row;
    // rule 1 (the other cell owns the face, or an interface) slot = slotBase[other] + k.
    bool compact = false;
    Table<uint32_t> entries16;        // [pair][lane] per slice
    Table<int32_t> sliceEntryStart16; // [nSlices+1], in words
    Table<uint16_t> slotBase;         // per tile: nc + nh + 1 values (the last one is the zero slot), padded to an even count
    Table<int32_t> tileSbStart;       // [nTiles+1] start of a tile's slotBase segment, in 32-bit words
    Table<int32_t> slotFace;        // caller face id; -1 = padding; <= -2 : interface slot -(2+ext)
    Table<int32_t> extSlot;         // [nExt] slot of each interface face
    Table<int32_t> interiorTiles, boundaryTiles;
    Table<int32_t> patchOffset;     // [nPatches+1]
    Table<int32_t> patchFaceCellsE; // [nExt] engine cell of each patch face
    Table<int32_t> faceSlot;        // [nFaces] a slot that holds caller face f (for faceH)
    int32_t maxCells = 0, maxSlots = 0, maxHalo = 0;
    int64_t totalSlots = 0;
};

struct TileParams {
    int32_t tileCells = 1024; // max cells per tile
    int32_t slotCap = 4094;   // max coefficient slots per tile
    bool compact = true;      // also build the 16-bit entry form when the mesh allows it
    // ORDERED layout: the caller's cell numbering is already tile-contiguous (a mesh renumbered with an earlier layout's
    // engine order, e.g. by renumberMesh) -- no clustering, engine order == caller order, e2c is the identity.
    //   keepOrder && givenTileStart: tile t = caller cells [givenTileStart[t], givenTileStart[t+1]), nGivenTiles tiles;
    //   keepOrder alone: consecutive cells are chunked greedily under the cell / slot caps.
    bool keepOrder = false;
    const int32_t* givenTileStart = nullptr;
    int32_t nGivenTiles = 0;
    // The clustering visits cells in index order, so its quality follows the locality of the caller's numbering: a mesh
    // numbered at random gets tiles of ~550 cells instead of 1024 and a 2.4x slower Amul.  reorder: -1 = when the mean
    // |upperAddr - lowerAddr| says the numbering has no locality (> 4 N^(2/3)), the clustering runs on a Cuthill-McKee
    // ordering of the cell graph instead (same tiles as for the well-numbered mesh); 0 = never; 1 = always.
    int32_t reorder = -1;
    // GIVEN partition (round 4; the GAMG level layouts that inherit their tiles, inherit_tiles below): givenPart[c] in
    // [0, nGivenParts) names the tile of caller cell c, every tile non-empty and within the caps; no clustering.  Tiles are
    // ordered by their smallest cell and their cells ascend, exactly as after the clustering -- the partition a clustered layout
    // has gives that layout's tables back.
    const int32_t* givenPart = nullptr;
    int32_t nGivenParts = 0;
};

// returns empty string on success, else an error message
std::string build_tile_layout(int32_t nCells, int32_t nFaces, const int32_t* lower,
                              const int32_t* upper, int32_t nPatches,
                              const int32_t* patchSizes, const int32_t* const* patchFaceCells,
                              const TileParams& prm, TileLayout& out,
                              const int32_t* const* patchNbrCells = nullptr);
// Tiles of a coarse GAMG level INHERITED from the tiles of its fine level instead of clustered from scratch (the clustering's
// sequential heavy-edge matching is what the level layouts of a hierarchy spend their time in): a coarse cell goes where its
// first child is (pair agglomeration halves a 1024-cell tile), then tiles are merged pairwise along their heaviest common
// boundary under the caps (tile graph of a few thousand vertices, in index order: deterministic).  On the levels of the 216^3
// box this gives the clustering's own tiles on the first two levels and +-0.6 % halo entries below (tools/exp/inherit_tiles.cpp).
// fineTileOfCell: tile of every fine cell in the fine level's CALLER numbering.  Returns an error text when a tile cannot be
// formed within the caps (the caller then clusters as before).
std::string inherit_tiles(int32_t nFine, const int32_t* restrictMap, const int32_t* fineTileOfCell, int32_t nFineTiles,
                          int32_t nCoarse, int32_t nCoarseFaces, const int32_t* cLower, const int32_t* cUpper,
                          int32_t nPatches, const int32_t* patchSizes, const int32_t* const* patchFaceCells,
                          int32_t cellCap, int32_t slotCap, Table<int32_t>& part, int32_t& nParts);
// patchNbrCells[p] != nullptr marks patch p as a LOCAL coupled patch (cyclic): face i couples faceCells[i]
// with the local cell patchNbrCells[p][i] (cyclicLduInterfaceField); nullptr = values arrive in the ext region.

} // namespace mi
