// gamg.cpp -- see gamg.hpp.
#include "gamg.hpp"
#include "host_parallel.hpp"

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

// Greedy pair matching of one level.  Visiting order and tie-breaking follow the reference
// (strict '>' => first maximum wins; a cell's faces are listed neighbour-side first, then
// owner-side; unmatched cells join the cluster across their heaviest face; leftovers become
// singletons; reverse sweeps mirror the coarse numbering).
int32_t match_pairs(int32_t nFine, int32_t nFaces, const int32_t* lower, const int32_t* upper,
                    const std::vector<double>& w, bool forward, std::vector<int32_t>& coarseOf)
{
    std::vector<int32_t> start((size_t)nFine + 1, 0);
    for (int32_t f = 0; f < nFaces; ++f) { ++start[(size_t)upper[f] + 1]; ++start[(size_t)lower[f] + 1]; }
    for (int32_t c = 0; c < nFine; ++c) start[(size_t)c + 1] += start[c];
    std::vector<int32_t> faces((size_t)2 * nFaces), fill(start.begin(), start.end() - 1);
    for (int32_t f = 0; f < nFaces; ++f) faces[(size_t)fill[upper[f]]++] = f;
    for (int32_t f = 0; f < nFaces; ++f) faces[(size_t)fill[lower[f]]++] = f;

    coarseOf.assign((size_t)nFine, -1);
    int32_t nCoarse = 0;
    const double NEG = -1e20;
    for (int32_t k = 0; k < nFine; ++k) {
        const int32_t c = forward ? k : nFine - 1 - k;
        if (coarseOf[c] >= 0) continue;
        int32_t pick = -1; double best = NEG;
        for (int32_t j = start[c]; j < start[(size_t)c + 1]; ++j) {
            const int32_t f = faces[j];
            if (coarseOf[upper[f]] < 0 && coarseOf[lower[f]] < 0 && w[f] > best) { pick = f; best = w[f]; }
        }
        if (pick >= 0) { coarseOf[upper[pick]] = coarseOf[lower[pick]] = nCoarse++; continue; }
        int32_t join = -1; double jbest = NEG;
        for (int32_t j = start[c]; j < start[(size_t)c + 1]; ++j) {
            const int32_t f = faces[j];
            if (w[f] > jbest) { join = f; jbest = w[f]; }
        }
        if (join >= 0) coarseOf[c] = std::max(coarseOf[upper[join]], coarseOf[lower[join]]);
    }
    for (int32_t k = 0; k < nFine; ++k) {
        const int32_t c = forward ? k : nFine - 1 - k;
        if (coarseOf[c] < 0) coarseOf[c] = nCoarse++;
    }
    if (!forward) for (int32_t c = 0; c < nFine; ++c) coarseOf[c] = nCoarse - 1 - coarseOf[c];
    return nCoarse;
}

// Coarse faces: one per unordered pair of distinct coarse cells, numbered so that they are
// grouped by owner (= smaller coarse cell) and, inside an owner, in order of first appearance
// among the fine faces -- the numbering the reference's per-cell neighbour lists produce.
void build_coarse_faces(GamgLevelHost& L, const int32_t* lower, const int32_t* upper)
{
    const int32_t nF = L.nFineFaces, nC = L.nCoarse;
    L.faceRestrict.assign((size_t)nF, 0);
    L.faceFlip.assign((size_t)nF, 0);
    // cut faces bucketed by owner (= smaller coarse cell), ascending fine face inside a bucket (counting sort is stable);
    // flat arrays and per-owner work only, so the owners are processed by the host threads independently
    std::vector<int32_t> oStart((size_t)nC + 1, 0);
    for (int32_t f = 0; f < nF; ++f) {
        const int32_t ru = L.restrictMap[upper[f]], rl = L.restrictMap[lower[f]];
        if (ru == rl) L.faceRestrict[f] = -(ru + 1);
        else ++oStart[(size_t)std::min(ru, rl) + 1];
    }
    for (int32_t c = 0; c < nC; ++c) oStart[(size_t)c + 1] += oStart[c];
    std::vector<int32_t> oFace((size_t)oStart[nC]), fill(oStart.begin(), oStart.end() - 1);
    for (int32_t f = 0; f < nF; ++f) if (L.faceRestrict[f] >= 0) oFace[(size_t)fill[std::min(L.restrictMap[upper[f]], L.restrictMap[lower[f]])]++] = f;
    // distinct neighbours of every owner in order of first appearance: first their number ...
    std::vector<int32_t> base((size_t)nC + 1, 0);
    auto nei_of = [&](int32_t f) { return std::max(L.restrictMap[upper[f]], L.restrictMap[lower[f]]); };
    parallel_blocks(nC, 32768, [&](int64_t b, int64_t e, int) {
        std::vector<int32_t> seen;
        for (int32_t c = (int32_t)b; c < (int32_t)e; ++c) {
            seen.clear();
            for (int32_t j = oStart[c]; j < oStart[(size_t)c + 1]; ++j) {
                const int32_t nei = nei_of(oFace[j]);
                if (std::find(seen.begin(), seen.end(), nei) == seen.end()) seen.push_back(nei);
            }
            base[(size_t)c + 1] = (int32_t)seen.size();
        }
    });
    for (int32_t c = 0; c < nC; ++c) base[(size_t)c + 1] += base[c];
    const int32_t nCF = base[nC];
    L.nCoarseFaces = nCF;
    L.cLower.resize(nCF); L.cUpper.resize(nCF);
    // ... then the numbering: coarse faces grouped by owner, inside an owner in order of first appearance among the fine faces
    parallel_blocks(nC, 32768, [&](int64_t b, int64_t e, int) {
        std::vector<int32_t> seen;
        for (int32_t c = (int32_t)b; c < (int32_t)e; ++c) {
            seen.clear();
            for (int32_t j = oStart[c]; j < oStart[(size_t)c + 1]; ++j) {
                const int32_t f = oFace[j], nei = nei_of(f);
                size_t at = std::find(seen.begin(), seen.end(), nei) - seen.begin();
                if (at == seen.size()) { seen.push_back(nei); L.cLower[(size_t)base[c] + at] = c; L.cUpper[(size_t)base[c] + at] = nei; }
                const int32_t t = base[c] + (int32_t)at;
                L.faceRestrict[f] = t;
                // flipped when the fine (lower -> upper) direction is opposite to the coarse (owner -> neighbour)
                if (c == L.restrictMap[upper[f]] && nei == L.restrictMap[lower[f]]) L.faceFlip[f] = 1;
            }
        }
    });
}

void segment(int32_t nTargets, const std::vector<int32_t>& target, std::vector<int32_t>& start, std::vector<int32_t>& child)
{
    // children of target t = all i with target[i] == t, ascending i (counting sort is stable); negatives skipped
    start.assign((size_t)nTargets + 1, 0);
    for (int32_t t : target) if (t >= 0) ++start[(size_t)t + 1];
    for (int32_t t = 0; t < nTargets; ++t) start[(size_t)t + 1] += start[t];
    child.resize((size_t)start[nTargets]);
    std::vector<int32_t> fill(start.begin(), start.end() - 1);
    for (int32_t i = 0; i < (int32_t)target.size(); ++i) if (target[i] >= 0) child[(size_t)fill[target[i]]++] = i;
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
    std::vector<double> w = faceWeights ? std::vector<double>(faceWeights, faceWeights + nFaces) : std::vector<double>((size_t)nFaces, 0.0);
    int32_t nFine = nCells, nF = nFaces;
    const int32_t *lo = lower, *up = upper;
    const int32_t nPatches = cpl ? cpl->nPatches : 0;
    std::vector<std::vector<int32_t>> pfc, pnb; // patch faceCells / local neighbour cells of the current fine level
    std::vector<GamgCoupling::Ami> pami;        // AMI tables of the current fine level (cyclicAMI patches)
    if (cpl) { pfc = cpl->faceCells; pnb = cpl->nbrCells; pami = cpl->ami; pami.resize((size_t)nPatches); }
    for (int32_t p = 0; p < nPatches; ++p) if (cpl->isLocal[p] == 2) {
        if (mergeLevels != 1) return "cyclicAMI patches are agglomerated with mergeLevels 1 only";
        const GamgCoupling::Ami& A = pami[(size_t)p];
        if (A.nbrPatch < 0 || A.nbrPatch >= nPatches || A.start.size() != pfc[p].size() + 1 || A.magSf.size() != pfc[p].size())
            return "cyclicAMI patch without complete AMI tables / face areas (mi_addr_set_ami_face_areas)";
        if (A.transport >= 0 && (A.transport >= nPatches || cpl->isLocal[(size_t)A.transport] != 0 || A.nPartner > (int32_t)pfc[(size_t)A.transport].size()))
            return "cyclicAMI patch with a partner on another rank: its transport patch must be a processor patch at least as large as the partner patch";
    }
    // partner on another rank: transport-patch face that carries the value behind partner face J, on the current fine level
    std::vector<std::vector<int32_t>> amiSrc((size_t)nPatches);
    for (int32_t p = 0; p < nPatches; ++p) if (cpl && cpl->isLocal[p] == 2 && pami[(size_t)p].transport >= 0) {
        amiSrc[(size_t)p].resize((size_t)pami[(size_t)p].nPartner);
        for (int32_t j = 0; j < pami[(size_t)p].nPartner; ++j) amiSrc[(size_t)p][(size_t)j] = j;
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
        { MI_TICK("coarse faces", H.levels.size()); build_coarse_faces(L, lo, up); }
        {
        MI_TICK("face weights", H.levels.size());
        std::vector<double> cw((size_t)L.nCoarseFaces, 0.0); // restrictFaceField (host): plain summation
        for (int32_t f = 0; f < nF; ++f) if (L.faceRestrict[f] >= 0) cw[L.faceRestrict[f]] += w[f];
        w.swap(cw);
        }
        if (nPatches > 0) {
            // coarse-cell ids on both sides of every coupled patch face
            std::vector<std::vector<int32_t>> mine((size_t)nPatches), theirs((size_t)nPatches), send((size_t)nPatches);
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
                std::vector<std::vector<int32_t>> recv((size_t)nPatches);
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
            std::vector<GamgCoupling::Ami> next((size_t)nPatches);
            for (int32_t p = 0; p < nPatches; ++p) if (cpl->isLocal[p] == 2) {
                const GamgCoupling::Ami& F = pami[(size_t)p];
                GamgPatchHost& P = L.patches[p];
                const std::vector<int32_t>& srcR = P.faceRestrict;
                std::vector<int32_t> remoteR, nextSrc;
                if (F.transport >= 0) {
                    // The partner's face restrict map, derived here: its coarse faces are its distinct coarse cells in order of
                    // first appearance over its faces (cyclicAMIGAMGInterface.C:47-165 -- what the partner rank builds for its
                    // own side), and the coarse cell behind partner face J is what the transport patch received for its face
                    // amiSrc[J] (theirs[transport]).  nextSrc: the coarse transport face that carries each new partner face.
                    const std::vector<int32_t>& th = theirs[(size_t)F.transport];
                    const std::vector<int32_t>& trR = L.patches[(size_t)F.transport].faceRestrict;
                    const std::vector<int32_t>& src = amiSrc[(size_t)p];
                    std::unordered_map<int32_t, int32_t> cellToFace;
                    remoteR.resize(src.size());
                    for (size_t J = 0; J < src.size(); ++J) {
                        const int32_t cc = th[(size_t)src[J]];
                        auto it = cellToFace.find(cc);
                        if (it == cellToFace.end()) {
                            const int32_t k = (int32_t)nextSrc.size();
                            cellToFace.emplace(cc, k);
                            nextSrc.push_back(trR[(size_t)src[J]]);
                            remoteR[J] = k;
                        } else remoteR[J] = it->second;
                    }
                }
                const std::vector<int32_t>& tgtR = F.transport >= 0 ? remoteR : L.patches[(size_t)F.nbrPatch].faceRestrict;
                const size_t nc = P.faceCells.size();
                std::vector<std::vector<int32_t>> el(nc);
                std::vector<std::vector<double>> wl(nc);
                P.amiMagSf.assign(nc, 0.0);
                for (size_t i = 0; i < srcR.size(); ++i) P.amiMagSf[(size_t)srcR[i]] += F.magSf[i];
                for (size_t i = 0; i < srcR.size(); ++i) {
                    std::vector<int32_t>& e = el[(size_t)srcR[i]];
                    std::vector<double>& ww = wl[(size_t)srcR[i]];
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
                N.transport = F.transport; N.nPartner = (int32_t)nextSrc.size();
                if (F.transport >= 0) { P.amiSrcFace = nextSrc; amiSrc[(size_t)p].swap(nextSrc); }
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
    std::vector<int32_t> interior((size_t)B.nFineFaces);
    for (int32_t f = 0; f < B.nFineFaces; ++f) interior[f] = B.faceRestrict[f] < 0 ? -1 - B.faceRestrict[f] : -1;
    segment(B.nCoarse, interior, B.diagChildStart, B.diagChild);
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

bool invert_dense(int n, std::vector<double>& A)
{
    std::vector<double> I((size_t)n * n, 0.0);
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
