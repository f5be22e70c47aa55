// gamg.hpp -- host-side GAMG hierarchy for the engine (built once per mesh and cached, as the
// reference caches its GAMGAgglomeration MeshObject: GAMGAgglomeration.C:132-182).
//
// Implements the reference's *algorithm* so that the same cells are agglomerated:
//   pair matching          pairGAMGAgglomerate.C:135-313 (greedy, max face weight, alternating sweep direction)
//   coarse addressing      GAMGAgglomerateLduAddressing.C:245-461 (coarse faces grouped by owner in creation order)
//   level loop / stop      pairGAMGAgglomerate.C:46-120 (mergeLevels: combineLevels), GAMGAgglomeration.C:72-81
// and derives the device tables the MI355X kernels need (segmented children lists instead of
// the reference's sort/target/targetStart triplets, GAMGAgglomerateLduAddressing.C:37-120).
#pragma once
#include "host_parallel.hpp"
#include <cstdint>
#include <functional>
#include <string>
#include <vector>

namespace mi {

// one coupled patch on one level (GAMGInterface, interfaces/GAMGInterface/GAMGInterface.H; processorGAMGInterface.C:54-128:
// one coarse interface face per distinct (local coarse cell, neighbour coarse cell) pair, in order of first
// appearance among the fine patch faces -- both sides visit matching faces in the same order, so they agree)
struct GamgPatchHost {
    Table<int32_t> faceCells;     // [nCoarseIfaceFaces] coarse cell on this side
    Table<int32_t> nbrCells;      // [nCoarseIfaceFaces] coarse cell on the other side (its owner's numbering)
    Table<int32_t> faceRestrict;  // [nFineIfaceFaces] -> coarse interface face
    Table<int32_t> childStart, child; // fine patch faces of every coarse interface face, ascending
    // cyclicAMI patch (cyclicAMIGAMGInterface.C:47-165): one coarse face per distinct LOCAL coarse cell; the AMI of the coarse
    // side is the fine one agglomerated over both sides' face maps (AMIInterpolation::agglomerate, AMIInterpolation.C:279-540)
    Table<int32_t> amiStart, amiAddr; // [nCoarse+1], coarse face of the neighbour patch
    Table<double> amiW, amiMagSf;     // weights normalised per coarse face; agglomerated face areas
    // cyclicAMI whose partner patch lives on another rank: amiAddr numbers the PARTNER's coarse faces (its own order: distinct
    // partner coarse cells in order of first appearance); amiSrcFace[J] = face of this level's TRANSPORT patch whose received
    // value is the coarse cell behind partner face J (what mi_addr_set_ami_patch_remote takes as address)
    Table<int32_t> amiSrcFace;
    // partner SIDE split over several ranks: amiSrcSlot[J] = which of the patch's transport patches carries it (all 0 for one partner),
    // amiPartCount[q] = partner coarse faces that arrive through transport q (partner faces are numbered piece by piece)
    Table<int32_t> amiSrcSlot, amiPartCount;
};

// (the per-face tables are written completely by threaded passes: a resize() that zeroes 100+ MB on one thread first is time)
template <class T> using HostVec = std::vector<T, NoInitAlloc<T>>;
struct GamgLevelHost {
    int32_t nFine = 0, nFineFaces = 0, nCoarse = 0, nCoarseFaces = 0;
    Table<int32_t> restrictMap;    // [nFine] -> coarse cell
    HostVec<int32_t> faceRestrict;       // [nFineFaces] coarse face or -(coarseCell+1)
    HostVec<uint8_t> faceFlip;           // [nFineFaces]
    HostVec<int32_t> cLower, cUpper;     // coarse addressing
    // segmented children (ascending fine index inside every segment = the reference's stable sort)
    Table<int32_t> cellChildStart, cellChild;   // children cells of every coarse cell
    Table<int32_t> faceChildStart, faceChild;   // fine faces mapped onto every coarse face
    Table<int32_t> diagChildStart, diagChild;   // fine faces interior to every coarse cell
    Table<GamgPatchHost> patches;               // coupled patches of the COARSE side of this level
};

// The two places where the ranks of a decomposed case have to talk while the hierarchy is built:
//   allAnd           continueAgglomerating (GAMGAgglomeration.C:72-81, reduce(andOp<bool>()))
//   nbrRestrict      the neighbour's coarse-cell ids of the patch-internal cells, patch face order
//                    (GAMGAgglomerateLduAddressing.C:464-520: initInternalFieldTransfer / internalFieldTransfer)
// Local (cyclic) patches are resolved by the builder itself; processor patches go through the callbacks.
struct GamgCoupling {
    int32_t nPatches = 0;
    Table<Table<int32_t>> faceCells;   // finest level, per patch
    Table<Table<int32_t>> nbrCells;    // finest level, per patch; empty vector = processor patch
    Table<char> isLocal;                     // per patch: 0 processor, 1 cyclic (nbrCells), 2 cyclicAMI (ami tables)
    // transport >= 0: the partner patch lives on another rank; `transport` is the processor patch of THIS domain that carries the
    // partner's patch-internal field (same size on both ranks), nPartner the partner patch's face count, addr numbers its faces
    // transports / partCount: the partner SIDE is split over several ranks -- one transport patch per piece, addr numbers the pieces' faces
    // concatenated (transport == transports[0], nPartner == the sum of partCount)
    struct Ami { int32_t nbrPatch = -1; Table<int32_t> start, addr; Table<double> w, magSf; int32_t transport = -1, nPartner = 0; Table<int32_t> transports, partCount; };
    Table<Ami> ami;                          // per patch (nbrPatch < 0: not an AMI patch); finest level
    bool (*allAnd)(void* user, bool v) = nullptr;
    // in: send[p] = local coarse ids of patch p's cells (processor patches only); out: recv[p] same length
    bool (*nbrRestrict)(void* user, int level, const Table<Table<int32_t>>& send, Table<Table<int32_t>>& recv) = nullptr;
    void* user = nullptr;
};

struct GamgHierarchyHost {
    Table<GamgLevelHost> levels;
    bool forwardOut = true;
    // Called (when set, mergeLevels == 1) as soon as a level's coarse addressing and patches are final -- BEFORE the pair
    // matching of the next level starts; levels is reserved up front, so &levels[level] stays valid.  The engine uses it to
    // build that level's tile layout and children lists on other threads while the sequential matching goes on.
    // The children lists (cellChild / faceChild / diagChild) of such a level are left to the callee: finish_gamg_level.
    std::function<void(int level)> onLevel;
};
// the segmented children lists of a level from its final maps (what build_gamg_hierarchy does for every level at its end)
void finish_gamg_level(GamgLevelHost& level);

// faceWeights: [nFaces] (faceAreaPair: |Sf/sqrt|Sf| o (1,1.01,1.02)|; algebraicPair: |upper|)
std::string build_gamg_hierarchy(int32_t nCells, int32_t nFaces, const int32_t* lower, const int32_t* upper,
                                 const double* faceWeights, int32_t nCellsInCoarsestLevel, bool forwardInit,
                                 GamgHierarchyHost& out, const GamgCoupling* coupling = nullptr, int32_t mergeLevels = 1,
                                 int32_t dummyLevels = 0);
// dummyLevels n > 0: dummyAgglomeration (GAMGAgglomerations/dummyAgglomeration/dummyAgglomeration.C:45-90) -- n levels whose
// restrict addressing is the identity (every level is the fine mesh again); face weights and nCellsInCoarsestLevel are unused.
// mergeLevels m > 1: every created level is m consecutive pair steps folded into one by
// GAMGAgglomeration::combineLevels (GAMGAgglomerateLduAddressing.C:606-760), including its face-flip rule (the flip
// of the LAST pair step is kept, the earlier ones are dropped -- the reference's behaviour, reproduced as is).

// dense inverse by Gauss-Jordan with partial pivoting (coarsest level); returns false if singular
bool invert_dense(int n, Table<double>& A);

} // namespace mi
