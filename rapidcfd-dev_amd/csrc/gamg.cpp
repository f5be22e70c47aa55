// gamg.cpp -- see gamg.hpp.
#include "gamg.hpp"
#include "host_parallel.hpp"
#include "host_match.hpp"

#include <algorithm>
#include <cmath>
#include <unordered_map>
#ifdef MI_TIMING
#include <chrono>
#include <cstdio>
namespace { struct GamgTick { const char* what; int level; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~GamgTick() { fprintf(stderr, "[gamg-host] level %2d %-28s %.4f s\n", level, what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); } }; }
#define MI_TICK(what, level) GamgTick tick_##__LINE__{what, (int)(level)}
#else
#define MI_TICK(what, level)
#endif

namespace mi {

namespace {

// faces of every cell in the reference's visiting order: neighbour side (upper == c) ascending, then owner side (lower == c)
// ascending.  Upper-triangular order (lduAddressing; the coarse levels built here as well) has the owners ascending, so the owner
// side is a RANGE of faces and needs no list (oF empty).
struct CellFaces {
    Table<int32_t> uS, oS; IndexList uF, oF;
    template <class Fn> void for_each_other(int32_t c, Fn fn) const { for (int32_t j = uS[(size_t)c]; j < uS[(size_t)c + 1]; ++j) fn(uF[(size_t)j]); }
    template <class Fn> void for_each_owned(int32_t c, Fn fn) const
    {
        if (oF.empty()) for (int32_t f = oS[(size_t)c]; f < oS[(size_t)c + 1]; ++f) fn(f);
        else for (int32_t j = oS[(size_t)c]; j < oS[(size_t)c + 1]; ++j) fn(oF[(size_t)j]);
    }
    template <class Fn> void for_each(int32_t c, Fn fn) const { for_each_other(c, fn); for_each_owned(c, fn); }
    int32_t degree(int32_t c) const { return uS[(size_t)c + 1] - uS[(size_t)c] + oS[(size_t)c + 1] - oS[(size_t)c]; }
};
// owner side as ranges (false: the owners are not ascending -- lists instead)
bool owner_ranges(int32_t nFine, int32_t nFaces, const int32_t* lower, Table<int32_t>& oS)
{
    std::atomic<bool> unsorted{false};
    parallel_blocks(nFaces, 1 << 18, [&](int64_t b, int64_t e, int) { bool u = false; for (int64_t f = std::max<int64_t>(b, 1); f < e; ++f) u |= lower[f] < lower[f - 1]; if (u) unsorted = true; });
    if (unsorted) return false;
    oS.resize((size_t)nFine + 1);
    parallel_for(nFaces, 1 << 18, [&](int64_t f) { for (int32_t c = f > 0 ? lower[f - 1] + 1 : 0; c <= lower[f]; ++c) oS[(size_t)c] = (int32_t)f; });
    for (int32_t c = nFaces > 0 ? lower[nFaces - 1] + 1 : 0; c <= nFine; ++c) oS[(size_t)c] = nFaces;
    return true;
}
void cell_faces(int32_t nFine, int32_t nFaces, const int32_t* lower, const int32_t* upper, CellFaces& F)
{
    bucket_items(nFaces, nFine, [&](int64_t f) { return upper[f]; }, F.uS, F.uF);
    F.oF.clear();
    if (!owner_ranges(nFine, nFaces, lower, F.oS)) bucket_items(nFaces, nFine, [&](int64_t f) { return lower[f]; }, F.oS, F.oF);
}

// A FORWARD sweep over a level whose owners are ascending, every weight choosable (see match_pairs_sequential: laterOnly): the
// candidates of a cell are across the faces it owns -- a range of faces, no list -- so the neighbour-side lists are not built at
// all.  They are only missed by the cells that find no free neighbour and JOIN across their heaviest face: those are collected,
// their neighbour-side faces bucketed in one pass over the faces, and the joins resolved in visiting order afterwards (a join
// reads the coarse cell of its target, which was paired, or joined at an earlier turn: the same value at either time).
int32_t match_pairs_forward_owned(int32_t nFine, int32_t nFaces, const int32_t* lower, const int32_t* upper, const Table<int32_t>& oS,
                                  const double* w, Table<int32_t>& coarseOf)
{
    (void)lower;
    coarseOf.assign((size_t)nFine, -1);
    int32_t nCoarse = 0;
    const double NEG = -1e20;
    Table<int32_t> joiners;
    for (int32_t c = 0; c < nFine; ++c) {
        if (coarseOf[c] >= 0) continue;
        int32_t pick = -1; double best = NEG;
        for (int32_t f = oS[(size_t)c]; f < oS[(size_t)c + 1]; ++f) if (w[f] > best && coarseOf[upper[f]] < 0) { pick = f; best = w[f]; }
        if (pick >= 0) coarseOf[upper[pick]] = coarseOf[c] = nCoarse++;
        else joiners.push_back(c);
    }
    if (!joiners.empty()) {
        Table<int32_t> jIndex((size_t)nFine, -1), jS; IndexList jF;
        for (size_t k = 0; k < joiners.size(); ++k) jIndex[(size_t)joiners[k]] = (int32_t)k;
        bucket_items(nFaces, (int32_t)joiners.size(), [&](int64_t f) { return jIndex[(size_t)upper[f]]; }, jS, jF);
        for (size_t k = 0; k < joiners.size(); ++k) {
            const int32_t c = joiners[k];
            int32_t join = -1; double jbest = NEG;
            for (int32_t j = jS[k]; j < jS[k + 1]; ++j) { const int32_t f = jF[(size_t)j]; if (w[f] > jbest) { join = f; jbest = w[f]; } }
            for (int32_t f = oS[(size_t)c]; f < oS[(size_t)c + 1]; ++f) if (w[f] > jbest) { join = f; jbest = w[f]; }
            if (join >= 0) coarseOf[c] = std::max(coarseOf[upper[join]], coarseOf[lower[join]]);
        }
    }
    for (int32_t c = 0; c < nFine; ++c) if (coarseOf[c] < 0) coarseOf[c] = nCoarse++;
    return nCoarse;
}

// the sequential loop (the default: see match_pairs).  laterOnly (every weight can be chosen: > -1e20, not NaN): a cell whose
// turn has passed belongs to a coarse cell -- paired, or joined across its heaviest face -- so only the neighbours whose turn
// is still to come can be free: on a forward sweep those are across the faces the cell OWNS, on a backward sweep across the
// others, and the other half of the list is skipped without a look at its state (same picks: it never held a free cell).
int32_t match_pairs_sequential(int32_t nFine, const int32_t* lower, const int32_t* upper, const CellFaces& F,
                               const double* w, bool forward, bool laterOnly, Table<int32_t>& coarseOf)
{
    coarseOf.assign((size_t)nFine, -1);
    int32_t nCoarse = 0;
    const double NEG = -1e20;
    for (int32_t k = 0; k < nFine; ++k) {
        const int32_t c = forward ? k : nFine - 1 - k;
        if (coarseOf[c] >= 0) continue;
        int32_t pick = -1; double best = NEG;
        if (laterOnly && forward) F.for_each_owned(c, [&](int32_t f) { if (w[f] > best && coarseOf[upper[f]] < 0) { pick = f; best = w[f]; } });
        else if (laterOnly) F.for_each_other(c, [&](int32_t f) { if (w[f] > best && coarseOf[lower[f]] < 0) { pick = f; best = w[f]; } });
        else
        F.for_each(c, [&](int32_t f) { if (coarseOf[upper[f]] < 0 && coarseOf[lower[f]] < 0 && w[f] > best) { pick = f; best = w[f]; } });
        if (pick >= 0) { coarseOf[upper[pick]] = coarseOf[lower[pick]] = nCoarse++; continue; }
        int32_t join = -1; double jbest = NEG;
        F.for_each(c, [&](int32_t f) { if (w[f] > jbest) { join = f; jbest = w[f]; } });
        if (join >= 0) coarseOf[c] = std::max(coarseOf[upper[join]], coarseOf[lower[join]]);
    }
    for (int32_t k = 0; k < nFine; ++k) {
        const int32_t c = forward ? k : nFine - 1 - k;
        if (coarseOf[c] < 0) coarseOf[c] = nCoarse++;
    }
    if (!forward) parallel_for(nFine, 1 << 18, [&](int64_t c) { coarseOf[(size_t)c] = nCoarse - 1 - coarseOf[(size_t)c]; });
    return nCoarse;
}

struct PairGraph {   // the cell graph of one level for greedy_match_parallel (host_match.hpp)
    typedef double Weight;
    static double none() { return -1e20; }
    int32_t n; bool forward;
    const int32_t *start, *nbr; const double* wj;   // per list entry: the cell across the face, the face's weight
    int64_t begin(int32_t v) const { return start[v]; }
    int64_t end(int32_t v) const { return start[(size_t)v + 1]; }
    int32_t other(int32_t, int64_t j) const { return nbr[j]; }
    bool better(int32_t, int64_t j, int32_t, double best) const { return wj[j] > best; }
    double weight(int32_t, int64_t j) const { return wj[j]; }
};

// Greedy pair matching of one level.  Visiting order and tie-breaking follow the reference
// (strict '>' => first maximum wins; a cell's faces are listed neighbour-side first, then
// owner-side; unmatched cells join the cluster across their heaviest face; leftovers become
// singletons; reverse sweeps mirror the coarse numbering).
// MI_MATCH_PARALLEL=1: large levels are matched by the host threads (greedy_match_parallel: the decisions of the sequential loop,
// taken as soon as what they depend on is known), the numbering follows from prefix counts in visiting order: a pair gets the rank of
// the cell whose turn made it, a cell that joined gets the coarse cell across its heaviest face (which that cell had at that
// moment: it was paired, or had joined at an earlier turn), singletons come last.
int32_t match_pairs(int32_t nFine, int32_t nFaces, const int32_t* lower, const int32_t* upper,
                    const double* w, bool forward, Table<int32_t>& coarseOf)
{
#ifdef MI_TIMING
    auto t__ = std::chrono::steady_clock::now();
    auto sub = [&](const char* what) { auto n = std::chrono::steady_clock::now(); if (nFaces > 4000000) fprintf(stderr, "[gamg-host]     %-30s %.4f s\n", what, std::chrono::duration<double>(n - t__).count()); t__ = n; };
#else
    auto sub = [](const char*) {};
#endif
    std::atomic<bool> odd{false};
    parallel_blocks(nFaces, 1 << 18, [&](int64_t b, int64_t e, int) { bool any = false; for (int64_t f = b; f < e; ++f) any |= !(w[(size_t)f] > -1e20); if (any) odd = true; });
    const bool parallel = !odd && host_threads() > 1 && nFine >= (1 << 15) && env_int_host("MI_MATCH_PARALLEL", 0) != 0;
    const bool laterOnly = !odd && env_int_host("MI_MATCH_LATER_ONLY", 1) != 0;
    sub("weights check");
    CellFaces F;
    if (laterOnly && forward && !parallel && owner_ranges(nFine, nFaces, lower, F.oS)) {
        const int32_t nc = match_pairs_forward_owned(nFine, nFaces, lower, upper, F.oS, w, coarseOf);
        sub("forward sweep over owned faces");
        return nc;
    }
    cell_faces(nFine, nFaces, lower, upper, F);
    sub("faces of every cell");
    // The parallel form (opt-in, MI_MATCH_PARALLEL=1: measured slower than the sequential loop on the 64-core host of the GPU
    // box, profiles/r04_p_*) relies on "a cell whose turn has passed is never free again", which holds when every face can be
    // chosen (weight > -1e20, not NaN: face areas and coefficient magnitudes are); anything else takes the sequential loop.
    if (!parallel) { const int32_t nc = match_pairs_sequential(nFine, lower, upper, F, w, forward, laterOnly, coarseOf); sub("the sequential loop"); return nc; }
    // per list entry: the cell across and the weight (the decisions are taken in wavefront order, not in index order: every
    // indirection less is a cache miss less)
    Table<int32_t> start((size_t)nFine + 1, 0);
    parallel_for(nFine, 1 << 18, [&](int64_t c) { start[(size_t)c + 1] = F.degree((int32_t)c); });
    parallel_inclusive_scan(start.data() + 1, (int64_t)nFine);
    Table<int32_t> nbr((size_t)start[(size_t)nFine]);
    Table<double> wj((size_t)start[(size_t)nFine]);
    parallel_for(nFine, 1 << 16, [&](int64_t c) {
        int32_t o = start[(size_t)c];
        F.for_each((int32_t)c, [&](int32_t f) { nbr[(size_t)o] = upper[f] == (int32_t)c ? lower[f] : upper[f]; wj[(size_t)o] = w[(size_t)f]; ++o; });
    });
    PairGraph g{nFine, forward, start.data(), nbr.data(), wj.data()};
    Table<int32_t> mate; Table<uint8_t> proposer;
    greedy_match_parallel(g, mate, proposer);
    // the cell across the heaviest face (first maximum in list order), or -1: where an unmatched cell joins
    auto join_target = [&](int32_t c) {
        int32_t join = -1; double jbest = -1e20;
        for (int32_t j = start[(size_t)c]; j < start[(size_t)c + 1]; ++j) if (wj[(size_t)j] > jbest) { join = nbr[(size_t)j]; jbest = wj[(size_t)j]; }
        return join;
    };
    // ranks in visiting order: pairs by the cell whose turn made them, then the singletons
    Table<int32_t> rank((size_t)nFine);
    auto cell_at = [&](int64_t k) { return forward ? (int32_t)k : nFine - 1 - (int32_t)k; };
    parallel_for(nFine, 1 << 18, [&](int64_t k) { rank[(size_t)k] = proposer[(size_t)cell_at(k)]; });
    parallel_inclusive_scan(rank.data(), (int64_t)nFine);
    const int32_t nPairs = nFine > 0 ? rank[(size_t)nFine - 1] : 0;
    Table<int32_t> single((size_t)nFine);
    parallel_for(nFine, 1 << 16, [&](int64_t k) { const int32_t c = cell_at(k); single[(size_t)k] = (mate[(size_t)c] == c && join_target(c) < 0) ? 1 : 0; });
    parallel_inclusive_scan(single.data(), (int64_t)nFine);
    const int32_t nCoarse = nPairs + (nFine > 0 ? single[(size_t)nFine - 1] : 0);
    coarseOf.resize((size_t)nFine);
    auto pos = [&](int32_t c) { return forward ? c : nFine - 1 - c; };
    parallel_for(nFine, 1 << 16, [&](int64_t cc) {
        int32_t c = (int32_t)cc, id;
        if (mate[(size_t)c] == c && join_target(c) < 0) id = nPairs + single[(size_t)pos(c)] - 1;
        else {
            while (mate[(size_t)c] == c) c = join_target(c);                       // joined: the cluster of the cell across the heaviest face
            const int32_t lead = proposer[(size_t)c] ? c : mate[(size_t)c];
            id = rank[(size_t)pos(lead)] - 1;
        }
        coarseOf[(size_t)cc] = forward ? id : nCoarse - 1 - id;
    });
    return nCoarse;
}

// Coarse faces: one per unordered pair of distinct coarse cells, numbered so that they are
// grouped by owner (= smaller coarse cell) and, inside an owner, in order of first appearance
// among the fine faces -- the numbering the reference's per-cell neighbour lists produce.
// w / cw: the face weights of the fine level and (filled here) of the coarse one: restrictFaceField, plain summation in ascending
// fine-face order per coarse face -- every coarse face belongs to one owner, whose faces are visited in that order.
void build_coarse_faces(GamgLevelHost& L, const int32_t* lower, const int32_t* upper, const double* w, HostVec<double>& cw)
{
    const int32_t nF = L.nFineFaces, nC = L.nCoarse;
#ifdef MI_TIMING
    auto t__ = std::chrono::steady_clock::now();
    auto sub = [&](const char* what) { auto n = std::chrono::steady_clock::now(); if (nF > 4000000) fprintf(stderr, "[gamg-host]     %-30s %.4f s\n", what, std::chrono::duration<double>(n - t__).count()); t__ = n; };
#else
    auto sub = [](const char*) {};
#endif
    L.faceRestrict.resize((size_t)nF);
    L.faceFlip.resize((size_t)nF);
    sub("allocate");
    // cut faces bucketed by owner (= smaller coarse cell), ascending fine face inside a bucket (bucket_items: the stable counting
    // sort, threaded); flat arrays and per-owner work only, so the owners are processed by the host threads independently
    Table<int32_t> oStart; IndexList oFace;
    parallel_for(nF, 1 << 16, [&](int64_t f) {   // interior: -(coarse cell + 1); cut: for now the owner
        const int32_t ru = L.restrictMap[(size_t)upper[f]], rl = L.restrictMap[(size_t)lower[f]];
        L.faceRestrict[(size_t)f] = ru == rl ? -(ru + 1) : std::min(ru, rl);
        L.faceFlip[(size_t)f] = 0;
    });
    sub("classify faces");
    bucket_items(nF, nC, [&](int64_t f) { return L.faceRestrict[(size_t)f] < 0 ? -1 : L.faceRestrict[(size_t)f]; }, oStart, oFace);
    sub("bucket cut faces by owner");
    // distinct neighbours of every owner in order of first appearance: per cut face its position in that list (and the flip:
    // the fine lower -> upper direction runs against the coarse owner -> neighbour one when the UPPER cell is in the owner) ...
    Table<int32_t> base((size_t)nC + 1, 0); IndexList oAt(oFace.size()), oNei(oFace.size());
    parallel_blocks(nC, 32768, [&](int64_t b, int64_t e, int) {
        Table<int32_t> seen;
        for (int32_t c = (int32_t)b; c < (int32_t)e; ++c) {
            seen.clear();
            for (int32_t j = oStart[c]; j < oStart[(size_t)c + 1]; ++j) {
                const int32_t f = oFace[(size_t)j];
                const int32_t ru = L.restrictMap[(size_t)upper[f]], rl = L.restrictMap[(size_t)lower[f]];
                const int32_t nei = std::max(ru, rl);
                const size_t at = std::find(seen.begin(), seen.end(), nei) - seen.begin();
                if (at == seen.size()) seen.push_back(nei);
                oAt[(size_t)j] = (int32_t)at; oNei[(size_t)j] = nei;
                if (ru == c) L.faceFlip[(size_t)f] = 1;
            }
            base[(size_t)c + 1] = (int32_t)seen.size();
        }
    });
    sub("distinct neighbours");
    parallel_inclusive_scan(base.data() + 1, (int64_t)nC);
    const int32_t nCF = base[nC];
    L.nCoarseFaces = nCF;
    L.cLower.resize(nCF); L.cUpper.resize(nCF);
    cw.resize((size_t)nCF);
    // ... then the numbering: coarse faces grouped by owner, inside an owner in order of first appearance among the fine faces
    parallel_blocks(nC, 32768, [&](int64_t b, int64_t e, int) {
        for (int32_t c = (int32_t)b; c < (int32_t)e; ++c) {
            int32_t known = 0;
            for (int32_t t = base[c]; t < base[(size_t)c + 1]; ++t) cw[(size_t)t] = 0.0;
            for (int32_t j = oStart[c]; j < oStart[(size_t)c + 1]; ++j) {
                const int32_t f = oFace[(size_t)j], t = base[c] + oAt[(size_t)j];
                if (oAt[(size_t)j] == known) { L.cLower[(size_t)t] = c; L.cUpper[(size_t)t] = oNei[(size_t)j]; ++known; }
                L.faceRestrict[(size_t)f] = t;
                cw[(size_t)t] += w[(size_t)f];
            }
        }
    });
    sub("numbering + weights");
    free_in_background(oStart, oFace, oAt, oNei, base);
}

template <class Vec>
void segment(int32_t nTargets, const Vec& target, Table<int32_t>& start, Table<int32_t>& child)
{
    // children of target t = all i with target[i] == t, ascending i (the stable counting sort, threaded); negatives skipped
    bucket_items((int64_t)target.size(), nTargets, [&](int64_t i) { return target[(size_t)i]; }, start, child);
}

} // namespace

std::string build_gamg_hierarchy(int32_t nCells, int32_t nFaces, const int32_t* lower, const int32_t* upper,
                                 const double* faceWeights, int32_t nCellsInCoarsestLevel, bool forwardInit,
                                 GamgHierarchyHost& H, const GamgCoupling* cpl, int32_t mergeLevels, int32_t dummyLevels)
{
    if (nCells <= 0 || (!faceWeights && dummyLevels <= 0) || mergeLevels < 1 || dummyLevels < 0) return "bad argument";
    if (dummyLevels > 0 && mergeLevels != 1) return "the dummy agglomeration has no mergeLevels";
    H.levels.clear();
    bool forward = forwardInit;
    const int maxLevels = 50;
    H.levels.reserve((size_t)maxLevels);          // element addresses stay valid for the onLevel consumers
    const bool pipelined = (bool)H.onLevel && mergeLevels == 1;
    // face weights of the current fine level: the caller's array on the finest level (no copy), then the restricted ones
    HostVec<double> wOwn;
    if (!faceWeights) wOwn.assign((size_t)nFaces, 0.0);
    const double* w = faceWeights ? faceWeights : wOwn.data();
    int32_t nFine = nCells, nF = nFaces;
    const int32_t *lo = lower, *up = upper;
    const int32_t nPatches = cpl ? cpl->nPatches : 0;
    Table<Table<int32_t>> pfc, pnb; // patch faceCells / local neighbour cells of the current fine level
    Table<GamgCoupling::Ami> pami;        // AMI tables of the current fine level (cyclicAMI patches)
    if (cpl) { pfc = cpl->faceCells; pnb = cpl->nbrCells; pami = cpl->ami; pami.resize((size_t)nPatches); }
    for (int32_t p = 0; p < nPatches; ++p) if (cpl->isLocal[p] == 2) {
        if (mergeLevels != 1) return "cyclicAMI patches are agglomerated with mergeLevels 1 only";
        const GamgCoupling::Ami& A = pami[(size_t)p];
        if (A.nbrPatch < 0 || A.nbrPatch >= nPatches || A.start.size() != pfc[p].size() + 1 || A.magSf.size() != pfc[p].size())
            return "cyclicAMI patch without complete AMI tables / face areas (mi_addr_set_ami_face_areas)";
        if (A.transport >= 0) {
            if (A.transports.empty()) { pami[(size_t)p].transports.assign(1, A.transport); pami[(size_t)p].partCount.assign(1, A.nPartner); }
            const GamgCoupling::Ami& B = pami[(size_t)p];
            for (size_t q = 0; q < B.transports.size(); ++q)
                if (B.transports[q] < 0 || B.transports[q] >= nPatches || cpl->isLocal[(size_t)B.transports[q]] != 0 || B.partCount[q] > (int32_t)pfc[(size_t)B.transports[q]].size())
                    return "cyclicAMI patch with a partner on another rank: its transport patches must be processor patches at least as large as the partner pieces";
        }
    }
    // partner on another rank: transport-patch face that carries the value behind partner face J, on the current fine level
    // (a partner side split over several ranks: partner faces numbered piece by piece, amiSlot = the piece = which transport patch)
    Table<Table<int32_t>> amiSrc((size_t)nPatches), amiSlot((size_t)nPatches);
    for (int32_t p = 0; p < nPatches; ++p) if (cpl && cpl->isLocal[p] == 2 && pami[(size_t)p].transport >= 0) {
        const GamgCoupling::Ami& A = pami[(size_t)p];
        for (size_t q = 0; q < A.transports.size(); ++q)
            for (int32_t j = 0; j < A.partCount[q]; ++j) { amiSrc[(size_t)p].push_back(j); amiSlot[(size_t)p].push_back((int32_t)q); }
    }
    int nPairLevels = 0;
    while ((int)H.levels.size() < maxLevels - 1) {
        GamgLevelHost L;
        L.nFine = nFine; L.nFineFaces = nF;
        bool cont;
        if (dummyLevels > 0) { // dummyAgglomeration.C:60-85: identity restrict addressing, nLevels of them
            L.restrictMap.resize((size_t)nFine);
            for (int32_t i = 0; i < nFine; ++i) L.restrictMap[i] = i;
            L.nCoarse = nFine;
            cont = (int32_t)H.levels.size() < dummyLevels;
        } else {
        { MI_TICK("pair matching", H.levels.size()); L.nCoarse = match_pairs(nFine, nF, lo, up, w, forward, L.restrictMap); }
        forward = !forward;
        cont = !(L.nCoarse < nCellsInCoarsestLevel || L.nCoarse == nFine); // continueAgglomerating
        }
        if (cpl && cpl->allAnd) cont = cpl->allAnd(cpl->user, cont);              // ... on all processors
        if (!cont) break;
        {
            MI_TICK("coarse faces + face weights", H.levels.size());
            HostVec<double> cw;   // restrictFaceField (host): plain summation, inside the per-owner pass
            build_coarse_faces(L, lo, up, w, cw);
            wOwn.swap(cw);
            w = wOwn.data();
        }
        if (nPatches > 0) {
            // coarse-cell ids on both sides of every coupled patch face
            Table<Table<int32_t>> mine((size_t)nPatches), theirs((size_t)nPatches), send((size_t)nPatches);
            bool anyRemote = false;
            for (int32_t p = 0; p < nPatches; ++p) {
                mine[p].resize(pfc[p].size());
                for (size_t i = 0; i < pfc[p].size(); ++i) mine[p][i] = L.restrictMap[pfc[p][i]];
                if (cpl->isLocal[p] == 2) continue; // cyclicAMI: no face-to-face partner
                if (cpl->isLocal[p]) {
                    theirs[p].resize(pnb[p].size());
                    for (size_t i = 0; i < pnb[p].size(); ++i) theirs[p][i] = L.restrictMap[pnb[p][i]];
                } else { send[p] = mine[p]; anyRemote = true; }
            }
            if (anyRemote) {
                if (!cpl->nbrRestrict) return "processor patches need the nbrRestrict callback";
                Table<Table<int32_t>> recv((size_t)nPatches);
                if (!cpl->nbrRestrict(cpl->user, (int)H.levels.size(), send, recv)) return "exchange of the restrict addressing failed";
                for (int32_t p = 0; p < nPatches; ++p) if (!cpl->isLocal[p]) {
                    if (recv[p].size() != mine[p].size()) return "exchange of the restrict addressing returned a wrong size";
                    theirs[p].swap(recv[p]);
                }
            }
            L.patches.resize((size_t)nPatches);
            for (int32_t p = 0; p < nPatches; ++p) if (cpl->isLocal[p] == 2) { // keyed by the local coarse cell alone, first appearance
                GamgPatchHost& P = L.patches[p];
                std::unordered_map<int32_t, int32_t> cellToFace;
                P.faceRestrict.resize(mine[p].size());
                for (size_t i = 0; i < mine[p].size(); ++i) {
                    auto it = cellToFace.find(mine[p][i]);
                    if (it == cellToFace.end()) {
                        const int32_t k = (int32_t)P.faceCells.size();
                        cellToFace.emplace(mine[p][i], k);
                        P.faceCells.push_back(mine[p][i]);
                        P.faceRestrict[i] = k;
                    } else P.faceRestrict[i] = it->second;
                }
                segment((int32_t)P.faceCells.size(), P.faceRestrict, P.childStart, P.child);
                pfc[p] = P.faceCells;
            }
            for (int32_t p = 0; p < nPatches; ++p) {
                if (cpl->isLocal[p] == 2) continue;
                GamgPatchHost& P = L.patches[p];
                std::unordered_map<uint64_t, int32_t> pairToFace;
                pairToFace.reserve(mine[p].size() * 2);
                P.faceRestrict.resize(mine[p].size());
                for (size_t i = 0; i < mine[p].size(); ++i) {
                    const uint64_t key = ((uint64_t)(uint32_t)mine[p][i] << 32) | (uint32_t)theirs[p][i];
                    auto it = pairToFace.find(key);
                    if (it == pairToFace.end()) {
                        const int32_t k = (int32_t)P.faceCells.size();
                        pairToFace.emplace(key, k);
                        P.faceCells.push_back(mine[p][i]); P.nbrCells.push_back(theirs[p][i]);
                        P.faceRestrict[i] = k;
                    } else P.faceRestrict[i] = it->second;
                }
                segment((int32_t)P.faceCells.size(), P.faceRestrict, P.childStart, P.child);
                pfc[p] = P.faceCells;
                if (cpl->isLocal[p]) pnb[p] = P.nbrCells;
            }
            // the coarse AMI of every cyclicAMI patch: fine faces in order, their addresses in order; an address whose coarse
            // target is already listed adds fineArea*weight, a new one is appended; then every list is divided by its sum
            // (normaliseWeights with conformal = true).  Host arithmetic of the reference: product, then addition.
            Table<GamgCoupling::Ami> next((size_t)nPatches);
            for (int32_t p = 0; p < nPatches; ++p) if (cpl->isLocal[p] == 2) {
                const GamgCoupling::Ami& F = pami[(size_t)p];
                GamgPatchHost& P = L.patches[p];
                const Table<int32_t>& srcR = P.faceRestrict;
                Table<int32_t> remoteR, nextSrc, nextSlot, nextCount;
                if (F.transport >= 0) {
                    // The partner's face restrict map, derived here: its coarse faces are its distinct coarse cells in order of
                    // first appearance over its faces (cyclicAMIGAMGInterface.C:47-165 -- what the partner rank builds for its
                    // own side), and the coarse cell behind partner face J is what the transport patch received for its face
                    // amiSrc[J] (theirs[transport]).  nextSrc: the coarse transport face that carries each new partner face.
                    // A partner side split over several ranks: every piece is agglomerated by ITS rank, so the first-appearance rule runs
                    // piece by piece (the pieces' faces are numbered one after the other, on every level).
                    const Table<int32_t>& src = amiSrc[(size_t)p];
                    const Table<int32_t>& slot = amiSlot[(size_t)p];
                    std::unordered_map<uint64_t, int32_t> cellToFace;
                    remoteR.resize(src.size());
                    nextCount.assign(F.transports.size(), 0);
                    for (size_t J = 0; J < src.size(); ++J) {
                        const int32_t q = slot[J], T = F.transports[(size_t)q];
                        const int32_t cc = theirs[(size_t)T][(size_t)src[J]];
                        const uint64_t key = ((uint64_t)(uint32_t)q << 32) | (uint32_t)cc;
                        auto it = cellToFace.find(key);
                        if (it == cellToFace.end()) {
                            const int32_t k = (int32_t)nextSrc.size();
                            cellToFace.emplace(key, k);
                            nextSrc.push_back(L.patches[(size_t)T].faceRestrict[(size_t)src[J]]);
                            nextSlot.push_back(q); ++nextCount[(size_t)q];
                            remoteR[J] = k;
                        } else remoteR[J] = it->second;
                    }
                }
                const Table<int32_t>& tgtR = F.transport >= 0 ? remoteR : L.patches[(size_t)F.nbrPatch].faceRestrict;
                const size_t nc = P.faceCells.size();
                Table<Table<int32_t>> el(nc);
                Table<Table<double>> wl(nc);
                P.amiMagSf.assign(nc, 0.0);
                for (size_t i = 0; i < srcR.size(); ++i) P.amiMagSf[(size_t)srcR[i]] += F.magSf[i];
                for (size_t i = 0; i < srcR.size(); ++i) {
                    Table<int32_t>& e = el[(size_t)srcR[i]];
                    Table<double>& ww = wl[(size_t)srcR[i]];
                    const double fineArea = F.magSf[i];
                    for (int32_t k = F.start[i]; k < F.start[i + 1]; ++k) {
                        const int32_t K = tgtR[(size_t)F.addr[(size_t)k]];
                        const double t = fineArea * F.w[(size_t)k];
                        size_t j = 0;
                        while (j < e.size() && e[j] != K) ++j;
                        if (j == e.size()) { e.push_back(K); ww.push_back(t); }
                        else ww[j] += t;
                    }
                }
                P.amiStart.assign(nc + 1, 0);
                for (size_t I = 0; I < nc; ++I) {
                    double sum = 0.0;
                    for (double v : wl[I]) sum += v;
                    for (size_t j = 0; j < el[I].size(); ++j) { P.amiAddr.push_back(el[I][j]); P.amiW.push_back(wl[I][j] / sum); }
                    P.amiStart[I + 1] = (int32_t)P.amiAddr.size();
                }
                GamgCoupling::Ami& N = next[(size_t)p];
                N.nbrPatch = F.nbrPatch; N.start = P.amiStart; N.addr = P.amiAddr; N.w = P.amiW; N.magSf = P.amiMagSf;
                N.transport = F.transport; N.nPartner = (int32_t)nextSrc.size(); N.transports = F.transports; N.partCount = nextCount;
                if (F.transport >= 0) { P.amiSrcFace = nextSrc; P.amiSrcSlot = nextSlot; P.amiPartCount = nextCount; amiSrc[(size_t)p].swap(nextSrc); amiSlot[(size_t)p].swap(nextSlot); }
            }
            for (int32_t p = 0; p < nPatches; ++p) if (cpl->isLocal[p] == 2) pami[(size_t)p] = std::move(next[(size_t)p]);
        }
        if (nPairLevels % mergeLevels) {
            // GAMGAgglomeration::combineLevels (GAMGAgglomerateLduAddressing.C:606-760): fold this pair step into the
            // previous level.  Face flips: the reference keeps the flip of this step only (:624-637).
            GamgLevelHost& P = H.levels.back();
            for (int32_t i = 0; i < P.nFineFaces; ++i) {
                const int32_t t = P.faceRestrict[i];
                if (t >= 0) { P.faceRestrict[i] = L.faceRestrict[t]; P.faceFlip[i] = L.faceFlip[t]; }
                else { P.faceRestrict[i] = -L.restrictMap[-t - 1] - 1; P.faceFlip[i] = 0; }
            }
            for (int32_t i = 0; i < P.nFine; ++i) P.restrictMap[i] = L.restrictMap[P.restrictMap[i]];
            P.nCoarse = L.nCoarse; P.nCoarseFaces = L.nCoarseFaces;
            P.cLower.swap(L.cLower); P.cUpper.swap(L.cUpper);
            for (int32_t p = 0; p < nPatches; ++p) {
                GamgPatchHost& PP = P.patches[p];
                const GamgPatchHost& LP = L.patches[p];
                for (size_t i = 0; i < PP.faceRestrict.size(); ++i) PP.faceRestrict[i] = LP.faceRestrict[PP.faceRestrict[i]];
                PP.faceCells = LP.faceCells; PP.nbrCells = LP.nbrCells;
                segment((int32_t)PP.faceCells.size(), PP.faceRestrict, PP.childStart, PP.child);
            }
        } else H.levels.push_back(std::move(L));
        if (pipelined) H.onLevel((int)H.levels.size() - 1);
        ++nPairLevels;
        GamgLevelHost& B = H.levels.back();
        nFine = B.nCoarse; nF = B.nCoarseFaces; lo = B.cLower.data(); up = B.cUpper.data();
    }
    MI_TICK("children lists (all levels)", -1);
    if (!pipelined) for (GamgLevelHost& B : H.levels) finish_gamg_level(B); // device tables from the final (possibly combined) maps
    H.forwardOut = forward;
    return std::string();
}

void finish_gamg_level(GamgLevelHost& B)
{
    segment(B.nCoarse, B.restrictMap, B.cellChildStart, B.cellChild);
    segment(B.nCoarseFaces, B.faceRestrict, B.faceChildStart, B.faceChild);
    bucket_items(B.nFineFaces, B.nCoarse, [&](int64_t f) { return B.faceRestrict[(size_t)f] < 0 ? -1 - B.faceRestrict[(size_t)f] : -1; }, B.diagChildStart, B.diagChild);
}

// Gauss-Jordan with partial pivoting on [A | I].  Row k of A is zero left of the pivot once the earlier columns are
// eliminated, so the update of A starts at column k (the skipped terms are x -= m*0: the same bits); the row updates are plain
// mul + sub (no contraction) and are compiled a second time for AVX2 -- four doubles per instruction, the same roundings --
// which the host of an MI355X box has (1.3 -> 0.5 ms at 151 cells: this runs once per coefficient binding of a GAMG solve).
template <int VARIANT>
static inline bool invert_dense_impl(int n, double* __restrict__ A, double* __restrict__ I)
{
    for (int k = 0; k < n; ++k) {
        int p = k; double big = std::fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; ++i) if (std::fabs(A[(size_t)i * n + k]) > big) { big = std::fabs(A[(size_t)i * n + k]); p = i; }
        if (big == 0.0) return false;
        if (p != k) for (int j = 0; j < n; ++j) { std::swap(A[(size_t)k * n + j], A[(size_t)p * n + j]); std::swap(I[(size_t)k * n + j], I[(size_t)p * n + j]); }
        const double inv = 1.0 / A[(size_t)k * n + k];
        double* __restrict__ Ak = A + (size_t)k * n; double* __restrict__ Ik = I + (size_t)k * n;
        for (int j = 0; j < n; ++j) { Ak[j] *= inv; Ik[j] *= inv; }
        for (int i = 0; i < n; ++i) {
            if (i == k) continue;
            double* __restrict__ Ai = A + (size_t)i * n; double* __restrict__ Ii = I + (size_t)i * n;
            const double m = Ai[k];
            if (m == 0.0) continue;
            for (int j = k; j < n; ++j) Ai[j] -= m * Ak[j];
            for (int j = 0; j < n; ++j) Ii[j] -= m * Ik[j];
        }
    }
    return true;
}
#if defined(__x86_64__)
__attribute__((target("avx2"))) static bool invert_dense_avx2(int n, double* A, double* I) { return invert_dense_impl<1>(n, A, I); }
#endif
static bool invert_dense_base(int n, double* A, double* I) { return invert_dense_impl<0>(n, A, I); }

bool invert_dense(int n, Table<double>& A)
{
    Table<double> I((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) I[(size_t)i * n + i] = 1.0;
    bool ok;
#if defined(__x86_64__)
    if (__builtin_cpu_supports("avx2")) ok = invert_dense_avx2(n, A.data(), I.data());
    else
#endif
    ok = invert_dense_base(n, A.data(), I.data());
    if (!ok) return false;
    A.swap(I);
    return true;
}

} // namespace mi
