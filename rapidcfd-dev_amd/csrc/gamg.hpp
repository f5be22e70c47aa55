// gamg.hpp -- host-side GAMG hierarchy for the engine (built once per mesh and cached, as the
// reference caches its GAMGAgglomeration MeshObject: GAMGAgglomeration.C:132-182).
//
// Implements the reference's *algorithm* so that the same cells are agglomerated:
//   pair matching          pairGAMGAgglomerate.C:135-313 (greedy, max face weight, alternating sweep direction)
//   coarse addressing      GAMGAgglomerateLduAddressing.C:245-461 (coarse faces grouped by owner in creation order)
//   level loop / stop      pairGAMGAgglomerate.C:46-120 (mergeLevels 1), GAMGAgglomeration.C:72-81
// and derives the device tables the MI355X kernels need (segmented children lists instead of
// the reference's sort/target/targetStart triplets, GAMGAgglomerateLduAddressing.C:37-120).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace mi {

struct GamgLevelHost {
    int32_t nFine = 0, nFineFaces = 0, nCoarse = 0, nCoarseFaces = 0;
    std::vector<int32_t> restrictMap;    // [nFine] -> coarse cell
    std::vector<int32_t> faceRestrict;   // [nFineFaces] coarse face or -(coarseCell+1)
    std::vector<uint8_t> faceFlip;       // [nFineFaces]
    std::vector<int32_t> cLower, cUpper; // coarse addressing
    // segmented children (ascending fine index inside every segment = the reference's stable sort)
    std::vector<int32_t> cellChildStart, cellChild;   // children cells of every coarse cell
    std::vector<int32_t> faceChildStart, faceChild;   // fine faces mapped onto every coarse face
    std::vector<int32_t> diagChildStart, diagChild;   // fine faces interior to every coarse cell
};

struct GamgHierarchyHost {
    std::vector<GamgLevelHost> levels;
    bool forwardOut = true;
};

// faceWeights: [nFaces] (faceAreaPair: |Sf/sqrt|Sf| o (1,1.01,1.02)|; algebraicPair: |upper|)
std::string build_gamg_hierarchy(int32_t nCells, int32_t nFaces, const int32_t* lower, const int32_t* upper,
                                 const double* faceWeights, int32_t nCellsInCoarsestLevel, bool forwardInit,
                                 GamgHierarchyHost& out);

// dense inverse by Gauss-Jordan with partial pivoting (coarsest level); returns false if singular
bool invert_dense(int n, std::vector<double>& A);

} // namespace mi
