/*
 * gamg_oracle.c -- CPU ORACLE for the GAMG solver (test infrastructure, NOT product code;
 * see the header of ldu_oracle.c: PARITY UNPINNED for the device arithmetic, the reference ships no tests).
 * PINNED BY THE REFERENCE'S OWN CODE: pair_agglomerate() below is checked, level by level, against
 * pairGAMGAgglomeration::agglomerate compiled from /root/reference (oracle/ref_shim, Makefile target `ref`,
 * tests/golden/golden_ref_pair.npz, tests/test_gamg.py::test_pair_agglomeration_equals_the_reference_code), and
 * orc_gamg_solve_sys() against the reference's GAMGSolverSolve.C (solve/Vcycle/initVcycle/solveCoarsestLevel compiled
 * the same way and run on the primitives below: tests/golden/golden_ref_gamg.npz,
 * tests/test_gamg.py::test_vcycle_equals_the_reference_source).
 *
 * Restates (single domain: orc_gamg_build/solve; coupled patches and decomposed cases: orc_gamg_build_sys/solve_sys
 * at the end of the file), paths relative to
 * /root/reference/src/OpenFOAM/matrices/lduMatrix/solvers/GAMG/ :
 *   - pair agglomeration        GAMGAgglomerations/pairGAMGAgglomeration/pairGAMGAgglomerate.C:31-313
 *   - coarse addressing         GAMGAgglomerations/GAMGAgglomeration/GAMGAgglomerateLduAddressing.C:245-461
 *   - face-weight restriction   GAMGAgglomerations/GAMGAgglomeration/GAMGAgglomerationTemplates.C:236-270
 *   - level loop / stop rule    pairGAMGAgglomerate.C:46-120, GAMGAgglomeration.C:72-81 (mergeLevels: combineLevels, GAMGAgglomerateLduAddressing.C:606-760)
 *   - coarse matrices           GAMGSolverAgglomerateMatrix.C:37-321 + GAMGSolverAgglomerateMatrixF.H:9-159
 *   - restrict / prolong        GAMGAgglomerationTemplates.C:35-153,273-308, GAMGAgglomerationF.H:9-38
 *   - scale                     GAMGSolverScale.C:40-171
 *   - V-cycle, solve loop       GAMGSolverSolve.C:59-474, defaults GAMGSolver.C:67-77
 *   - coarsest level            GAMGSolverSolve.C:552-570 (direct solve; dense LU with partial pivoting
 *                               stands in for LUscalarMatrix, src/OpenFOAM/matrices/LUscalarMatrix)
 *   - smoother                  Jacobi omega 0.9 (= the reference's "GaussSeidel"), via ldu_oracle.c
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ldu_oracle.h"

/* from ldu_oracle.c */
orc_system *orc_sys_create(int nDomains);
void orc_sys_set_iface_ami(orc_system *s, int d, int p, const label *start, const label *addr, const scalar *w, const unsigned char *low);
void orc_sys_set_iface_ami_parts(orc_system *s, int d, int p, int nParts, const label *partDomain, const label *partPatch);
void orc_sys_set_iface_transform(orc_system *s, int d, int p, scalar factor);
int orc_sys_add_interface(orc_system *s, int d, label nbrDomain, label nbrPatch, label nFaces, const label *faceCells,
                          const scalar *bouCoeffs, const scalar *intCoeffs);
void orc_sys_set_domain(orc_system *s, int d, label nCells, label nFaces, const label *lower, const label *upper,
                        const scalar *diag, const scalar *lowerC, const scalar *upperC);
void orc_sys_destroy(orc_system *s);
void orc_amul(const orc_system *s, const scalar *psi, scalar *Apsi);
void orc_jacobi_smooth(const orc_system *s, scalar omega, scalar *psi, const scalar *source, int nSweeps);
scalar orc_norm_factor(const orc_system *s, const scalar *psi, const scalar *source, const scalar *Apsi, scalar *tmp);
scalar orc_gSumMag(const orc_system *s, const scalar *a);
/* the Krylov solvers of ldu_oracle.c (their control / performance records, ldu_oracle.c:669-678) */
typedef struct { scalar initialResidual, finalResidual, normFactor; int32_t nIterations, converged, singular; } orc_perf;
typedef struct { scalar tolerance, relTol; int32_t maxIter, minIter; } orc_controls;
void orc_pcg_solve(const orc_system *s, scalar *psi, const scalar *source, const orc_controls *ctl, int precondKind, orc_perf *perf, scalar *hist, int histLen);
void orc_pbicg_solve(const orc_system *s, scalar *psi, const scalar *source, const orc_controls *ctl, int precondKind, orc_perf *perf, scalar *hist, int histLen);


#define G_GREAT 1e20
#define G_SMALL 1e-20
#define G_VSMALL 1e-300

typedef struct {
    label nFine, nFineFaces, nCoarse, nCoarseFaces;
    label *restrictMap;   /* [nFine] coarse cell of each fine cell */
    label *faceRestrict;  /* [nFineFaces] coarse face, or -(coarseCell+1) for faces inside a coarse cell */
    unsigned char *faceFlip;
    label *cLower, *cUpper; /* coarse owner / neighbour, grouped by owner in creation order */
} gamg_level;

typedef struct {
    int nLevels;          /* number of coarse levels */
    gamg_level *lev;
    int forwardOut;       /* value of the static sweep-direction flag after the build */
} gamg_hier;

/* one level of pair matching: pairGAMGAgglomerate.C:135-313 */
static label pair_agglomerate(label nFine, label nFaces, const label *lower, const label *upper,
                              const scalar *w, int forward, label *map)
{
    label *off = (label *)calloc((size_t)nFine + 1, sizeof(label));
    label *cf = (label *)malloc(sizeof(label) * (size_t)(2 * nFaces + 1));
    label *cnt = (label *)calloc((size_t)nFine, sizeof(label));
    label f, c, nCoarse = 0;
    for (f = 0; f < nFaces; f++) { off[upper[f] + 1]++; off[lower[f] + 1]++; }
    for (c = 0; c < nFine; c++) off[c + 1] += off[c];
    /* faces where the cell is the neighbour first, then faces where it is the owner (:176-193) */
    for (f = 0; f < nFaces; f++) cf[off[upper[f]] + cnt[upper[f]]++] = f;
    for (f = 0; f < nFaces; f++) cf[off[lower[f]] + cnt[lower[f]]++] = f;
    for (c = 0; c < nFine; c++) map[c] = -1;
    for (label k = 0; k < nFine; k++) {
        c = forward ? k : nFine - k - 1;
        if (map[c] >= 0) continue;
        label match = -1; scalar best = -G_GREAT;
        for (label j = off[c]; j < off[c + 1]; j++) {
            f = cf[j];
            if (map[upper[f]] < 0 && map[lower[f]] < 0 && w[f] > best) { match = f; best = w[f]; }
        }
        if (match >= 0) { map[upper[match]] = nCoarse; map[lower[match]] = nCoarse; nCoarse++; }
        else {
            label cm = -1; scalar cb = -G_GREAT;
            for (label j = off[c]; j < off[c + 1]; j++) { f = cf[j]; if (w[f] > cb) { cm = f; cb = w[f]; } }
            if (cm >= 0) { label a = map[upper[cm]], b = map[lower[cm]]; map[c] = a > b ? a : b; }
        }
    }
    for (label k = 0; k < nFine; k++) { c = forward ? k : nFine - k - 1; if (map[c] < 0) map[c] = nCoarse++; }
    if (!forward) { nCoarse--; for (c = 0; c < nFine; c++) map[c] = nCoarse - map[c]; nCoarse++; }
    free(off); free(cf); free(cnt);
    return nCoarse;
}

/* GAMGAgglomerateLduAddressing.C:245-461 */
static void coarse_addressing(gamg_level *L, const label *lower, const label *upper)
{
    const label nF = L->nFineFaces, nC = L->nCoarse;
    const label *rm = L->restrictMap;
    label maxN = 10, f, nCF = 0;
    label *ccn = (label *)calloc((size_t)nC, sizeof(label));
    label *ccf = (label *)malloc(sizeof(label) * (size_t)maxN * (size_t)nC);
    label *initNei = (label *)malloc(sizeof(label) * (size_t)(nF ? nF : 1));
    L->faceRestrict = (label *)malloc(sizeof(label) * (size_t)(nF ? nF : 1));
    L->faceFlip = (unsigned char *)calloc((size_t)(nF ? nF : 1), 1);
    for (f = 0; f < nF; f++) {
        label ru = rm[upper[f]], rl = rm[lower[f]];
        if (ru == rl) { L->faceRestrict[f] = -(ru + 1); continue; }
        label cOwn = ru < rl ? ru : rl, cNei = ru < rl ? rl : ru;
        int found = 0;
        for (label i = 0; i < ccn[cOwn]; i++)
            if (initNei[ccf[(size_t)maxN * cOwn + i]] == cNei) { found = 1; L->faceRestrict[f] = ccf[(size_t)maxN * cOwn + i]; break; }
        if (!found) {
            if (ccn[cOwn] >= maxN) {
                label oldMax = maxN; maxN *= 2;
                ccf = (label *)realloc(ccf, sizeof(label) * (size_t)maxN * (size_t)nC);
                for (label i = nC - 1; i >= 0; i--)
                    for (label j = ccn[i] - 1; j >= 0; j--) ccf[(size_t)maxN * i + j] = ccf[(size_t)oldMax * i + j];
            }
            ccf[(size_t)maxN * cOwn + ccn[cOwn]++] = nCF;
            initNei[nCF] = cNei;
            L->faceRestrict[f] = nCF++;
        }
    }
    L->nCoarseFaces = nCF;
    L->cLower = (label *)malloc(sizeof(label) * (size_t)(nCF ? nCF : 1));
    L->cUpper = (label *)malloc(sizeof(label) * (size_t)(nCF ? nCF : 1));
    label *cmap = (label *)malloc(sizeof(label) * (size_t)(nCF ? nCF : 1));
    label k = 0;
    for (label c = 0; c < nC; c++)
        for (label i = 0; i < ccn[c]; i++) {
            label init = ccf[(size_t)maxN * c + i];
            L->cLower[k] = c; L->cUpper[k] = initNei[init]; cmap[init] = k++;
        }
    for (f = 0; f < nF; f++) if (L->faceRestrict[f] >= 0) L->faceRestrict[f] = cmap[L->faceRestrict[f]];
    for (f = 0; f < nF; f++) {
        label cfi = L->faceRestrict[f];
        if (cfi >= 0 && L->cLower[cfi] == rm[upper[f]] && L->cUpper[cfi] == rm[lower[f]]) L->faceFlip[f] = 1;
    }
    free(ccn); free(ccf); free(initNei); free(cmap);
}

/* GAMGAgglomeration::combineLevels (GAMGAgglomerateLduAddressing.C:606-760): fold pair step C into the previous level P
 * (mergeLevels > 1, pairGAMGAgglomerate.C:110-117).  Restrict maps are composed; a face that was a coarse face of P
 * takes C's target AND C's flip -- the flip P recorded is dropped (:624-629), exactly as the reference does; a face
 * already inside a coarse cell of P follows that cell (:631-636; its flip is never read).  P takes C's coarse mesh. */
static void combine_levels(gamg_level *P, gamg_level *C)
{
    label i;
    for (i = 0; i < P->nFineFaces; i++) {
        const label t = P->faceRestrict[i];
        if (t >= 0) { P->faceRestrict[i] = C->faceRestrict[t]; P->faceFlip[i] = C->faceFlip[t]; }
        else { P->faceRestrict[i] = -C->restrictMap[-t - 1] - 1; P->faceFlip[i] = 0; }
    }
    for (i = 0; i < P->nFine; i++) P->restrictMap[i] = C->restrictMap[P->restrictMap[i]];
    P->nCoarse = C->nCoarse; P->nCoarseFaces = C->nCoarseFaces;
    free(P->cLower); free(P->cUpper);
    P->cLower = C->cLower; P->cUpper = C->cUpper;
    free(C->restrictMap); free(C->faceRestrict); free(C->faceFlip);
    memset(C, 0, sizeof(*C));
}

gamg_hier *orc_gamg_build_merged(label nCells, label nFaces, const label *lower, const label *upper,
                                 const scalar *faceWeights, label nCellsInCoarsestLevel, int forwardInit, int mergeLevels);
gamg_hier *orc_gamg_build(label nCells, label nFaces, const label *lower, const label *upper,
                          const scalar *faceWeights, label nCellsInCoarsestLevel, int forwardInit)
{
    return orc_gamg_build_merged(nCells, nFaces, lower, upper, faceWeights, nCellsInCoarsestLevel, forwardInit, 1);
}

gamg_hier *orc_gamg_build_merged(label nCells, label nFaces, const label *lower, const label *upper,
                                 const scalar *faceWeights, label nCellsInCoarsestLevel, int forwardInit, int mergeLevels)
{
    const int maxLevels = 50; /* GAMGAgglomeration.C:94 */
    gamg_hier *H = (gamg_hier *)calloc(1, sizeof(gamg_hier));
    H->lev = (gamg_level *)calloc((size_t)maxLevels, sizeof(gamg_level));
    int forward = forwardInit;
    label nFine = nCells, nF = nFaces;
    const label *lo = lower, *up = upper;
    scalar *w = (scalar *)malloc(sizeof(scalar) * (size_t)(nF ? nF : 1));
    memcpy(w, faceWeights, sizeof(scalar) * (size_t)nF);
    int nPairLevels = 0;
    while (H->nLevels < maxLevels - 1) {
        gamg_level *L = &H->lev[H->nLevels];
        L->nFine = nFine; L->nFineFaces = nF;
        L->restrictMap = (label *)malloc(sizeof(label) * (size_t)nFine);
        L->nCoarse = pair_agglomerate(nFine, nF, lo, up, w, forward, L->restrictMap);
        forward = !forward; /* the static flag flips even when the level is then discarded (:310) */
        if (!(L->nCoarse >= nCellsInCoarsestLevel) || L->nCoarse == nFine) { free(L->restrictMap); L->restrictMap = NULL; break; }
        coarse_addressing(L, lo, up);
        scalar *cw = (scalar *)calloc((size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1), sizeof(scalar));
        for (label f = 0; f < nF; f++) if (L->faceRestrict[f] >= 0) cw[L->faceRestrict[f]] += w[f];
        free(w); w = cw;
        if (nPairLevels % mergeLevels) { combine_levels(&H->lev[H->nLevels - 1], L); L = &H->lev[H->nLevels - 1]; }
        else H->nLevels++;
        nPairLevels++;
        nFine = L->nCoarse; nF = L->nCoarseFaces; lo = L->cLower; up = L->cUpper;
    }
    free(w);
    H->forwardOut = forward;
    return H;
}

/* dummyAgglomeration (GAMGAgglomerations/dummyAgglomeration/dummyAgglomeration.C:45-90): nLevels levels whose restrict
 * addressing is the identity, agglomerateLduAddressing run on each */
gamg_hier *orc_gamg_build_dummy(label nCells, label nFaces, const label *lower, const label *upper, int nLevels)
{
    gamg_hier *H = (gamg_hier *)calloc(1, sizeof(gamg_hier));
    H->lev = (gamg_level *)calloc((size_t)(nLevels + 1), sizeof(gamg_level));
    label nF = nFaces;
    const label *lo = lower, *up = upper;
    for (int l = 0; l < nLevels; l++) {
        gamg_level *L = &H->lev[l];
        L->nFine = nCells; L->nFineFaces = nF;
        L->restrictMap = (label *)malloc(sizeof(label) * (size_t)nCells);
        for (label i = 0; i < nCells; i++) L->restrictMap[i] = i;
        L->nCoarse = nCells;
        coarse_addressing(L, lo, up);
        nF = L->nCoarseFaces; lo = L->cLower; up = L->cUpper;
        H->nLevels++;
    }
    H->forwardOut = 1;
    return H;
}

int orc_gamg_n_levels(const gamg_hier *H) { return H->nLevels; }
int orc_gamg_forward_out(const gamg_hier *H) { return H->forwardOut; }
void orc_gamg_level_sizes(const gamg_hier *H, int l, label *out4)
{
    out4[0] = H->lev[l].nFine; out4[1] = H->lev[l].nFineFaces; out4[2] = H->lev[l].nCoarse; out4[3] = H->lev[l].nCoarseFaces;
}
void orc_gamg_level_maps(const gamg_hier *H, int l, label *restrictMap, label *faceRestrict, label *faceFlip,
                         label *cLower, label *cUpper)
{
    const gamg_level *L = &H->lev[l];
    label i;
    if (restrictMap) memcpy(restrictMap, L->restrictMap, sizeof(label) * (size_t)L->nFine);
    if (faceRestrict) memcpy(faceRestrict, L->faceRestrict, sizeof(label) * (size_t)L->nFineFaces);
    if (faceFlip) for (i = 0; i < L->nFineFaces; i++) faceFlip[i] = L->faceFlip[i];
    if (cLower) memcpy(cLower, L->cLower, sizeof(label) * (size_t)L->nCoarseFaces);
    if (cUpper) memcpy(cUpper, L->cUpper, sizeof(label) * (size_t)L->nCoarseFaces);
}
void orc_gamg_free(gamg_hier *H)
{
    for (int l = 0; l < H->nLevels; l++) {
        free(H->lev[l].restrictMap); free(H->lev[l].faceRestrict); free(H->lev[l].faceFlip);
        free(H->lev[l].cLower); free(H->lev[l].cUpper);
    }
    free(H->lev); free(H);
}

/* ---- fields between levels ---------------------------------------------------------------- */
/* restrictField: cf = 0; cf[c] = sum of ff over the children of c in ascending fine index
 * (segmented sum over the stable sort, GAMGAgglomerationF.H:9-38)                              */
static void restrict_field(const gamg_level *L, const scalar *ff, scalar *cf)
{
    label i;
    for (i = 0; i < L->nCoarse; i++) cf[i] = 0;
    for (i = 0; i < L->nFine; i++) cf[L->restrictMap[i]] += ff[i];
}
static void prolong_field(const gamg_level *L, const scalar *cf, scalar *ff)
{
    for (label i = 0; i < L->nFine; i++) ff[i] = cf[L->restrictMap[i]];
}

void orc_gamg_restrict_level(const gamg_hier *H, int l, const scalar *ff, scalar *cf) { restrict_field(&H->lev[l], ff, cf); }
void orc_gamg_prolong_level(const gamg_hier *H, int l, const scalar *cf, scalar *ff) { prolong_field(&H->lev[l], cf, ff); }

/* coarse matrix: GAMGSolverAgglomerateMatrix.C:65-72,218-317 (+F.H).  Sums run over the fine
 * faces in ascending index (stable sort by target).                                            */
static void agglomerate_matrix(const gamg_level *L, int asym, const scalar *fDiag, const scalar *fUpper,
                               const scalar *fLower, scalar *cDiag, scalar *cUpper, scalar *cLower)
{
    label f;
    restrict_field(L, fDiag, cDiag);
    for (f = 0; f < L->nCoarseFaces; f++) { cUpper[f] = 0; if (asym) cLower[f] = 0; }
    for (f = 0; f < L->nFineFaces; f++) {
        label t = L->faceRestrict[f];
        if (t >= 0) {
            if (!asym) cUpper[t] += fUpper[f];
            else if (!L->faceFlip[f]) { cUpper[t] += fUpper[f]; cLower[t] += fLower[f]; }
            else { cUpper[t] += fLower[f]; cLower[t] += fUpper[f]; }
        } else {
            if (!asym) cDiag[-1 - t] += 2 * fUpper[f];
            else cDiag[-1 - t] += fUpper[f] + fLower[f];
        }
    }
}

/* dense LU with partial pivoting for the coarsest level */
static void lu_factor(int n, scalar *A, int *piv)
{
    for (int k = 0; k < n; k++) {
        int p = k; scalar big = fabs(A[(size_t)k * n + k]);
        for (int i = k + 1; i < n; i++) if (fabs(A[(size_t)i * n + k]) > big) { big = fabs(A[(size_t)i * n + k]); p = i; }
        piv[k] = p;
        if (p != k) for (int j = 0; j < n; j++) { scalar t = A[(size_t)k * n + j]; A[(size_t)k * n + j] = A[(size_t)p * n + j]; A[(size_t)p * n + j] = t; }
        for (int i = k + 1; i < n; i++) {
            scalar m = A[(size_t)i * n + k] / A[(size_t)k * n + k];
            A[(size_t)i * n + k] = m;
            for (int j = k + 1; j < n; j++) A[(size_t)i * n + j] -= m * A[(size_t)k * n + j];
        }
    }
}
static void lu_solve(int n, const scalar *A, const int *piv, scalar *b)
{
    for (int k = 0; k < n; k++) { if (piv[k] != k) { scalar t = b[k]; b[k] = b[piv[k]]; b[piv[k]] = t; }
        for (int i = k + 1; i < n; i++) b[i] -= A[(size_t)i * n + k] * b[k]; }
    for (int k = n - 1; k >= 0; k--) { for (int j = k + 1; j < n; j++) b[k] -= A[(size_t)k * n + j] * b[j]; b[k] /= A[(size_t)k * n + k]; }
}

typedef struct {
    scalar tolerance, relTol;
    int32_t maxIter, minIter;
    int32_t nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps;
    int32_t nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps;
    int32_t nFinestSweeps, scaleCorrection; /* scaleCorrection < 0: default = symmetric (GAMGSolver.C:76) */
    scalar omega; /* Jacobi relaxation, JacobiSmoother.C:34-36 */
    int32_t directSolveCoarsest, reserved; /* GAMGSolver.C:77,232: default true; false = ICCG / BICCG on the coarsest level */
} gamg_controls;

typedef struct {
    scalar initialResidual, finalResidual, normFactor;
    int32_t nIterations, converged, singular;
} gamg_perf;

/* scale: GAMGSolverScale.C:59-171 */
static void scale_field(const orc_system *A, const scalar *D, label n, scalar *field, scalar *Acf, const scalar *source)
{
    orc_amul(A, field, Acf);
    long double num = 0, den = 0;
    for (label i = 0; i < n; i++) { num += (long double)source[i] * field[i]; den += (long double)Acf[i] * field[i]; }
    scalar d = (scalar)den;
    scalar sf = (scalar)num / (d >= 0 ? d + G_VSMALL : d - G_VSMALL); /* stabilise(), Scalar.H:295-305 */
    for (label i = 0; i < n; i++) field[i] = fma(sf, field[i], fma(-sf, Acf[i], source[i]) / D[i]);
}

/* solveCoarsestLevel without the direct solver (GAMGSolverSolve.C:572-613): ICCG (= PCG + DIC, here AINV) on a symmetric,
 * BICCG (= PBiCG + DILU) on an asymmetric coarsest matrix, zero initial guess, GAMG's own tolerance and relTol
 * (ICCG.C solverDict), default maxIter 1000 */
static void coarsest_iterative(const orc_system *Ac, int asym, const gamg_controls *ctl, const scalar *src, scalar *corr)
{
    orc_controls c; orc_perf p;
    memset(&c, 0, sizeof(c));
    c.tolerance = ctl->tolerance; c.relTol = ctl->relTol; c.maxIter = 1000; c.minIter = 0;
    memset(corr, 0, sizeof(scalar) * (size_t)Ac->nTotal);
    if (asym) orc_pbicg_solve(Ac, corr, src, &c, 2, &p, NULL, 0);
    else orc_pcg_solve(Ac, corr, src, &c, 2, &p, NULL, 0);
}

static int conv_check(gamg_perf *p, const gamg_controls *c)
{
    p->converged = (p->finalResidual < c->tolerance) || (c->relTol > G_SMALL && p->finalResidual < c->relTol * p->initialResidual);
    return p->converged;
}

static int imin(int a, int b) { return a < b ? a : b; }

void orc_gamg_solve(const gamg_hier *H, const label *lower, const label *upper, const scalar *diag,
                    const scalar *upperC, const scalar *lowerC /* NULL: symmetric */, scalar *psi,
                    const scalar *source, const gamg_controls *ctl, gamg_perf *perf, scalar *hist, int histLen)
{
    const int nL = H->nLevels;
    const int asym = lowerC != NULL;
    const int doScale = ctl->scaleCorrection < 0 ? !asym : ctl->scaleCorrection;
    const label n0 = H->lev[0].nFine;
    memset(perf, 0, sizeof(*perf));
    if (nL < 1) { perf->singular = 1; return; } /* reference: FatalError "No coarse levels created" */

    /* ---- level matrices (GAMGSolver.C:88-97: rebuilt on every solver construction) ---- */
    orc_system **A = (orc_system **)calloc((size_t)nL + 1, sizeof(*A));
    scalar **D = (scalar **)calloc((size_t)nL + 1, sizeof(*D));
    scalar **U = (scalar **)calloc((size_t)nL + 1, sizeof(*U));
    scalar **Lw = (scalar **)calloc((size_t)nL + 1, sizeof(*Lw));
    A[0] = orc_sys_create(1);
    orc_sys_set_domain(A[0], 0, n0, H->lev[0].nFineFaces, lower, upper, diag, lowerC, upperC);
    D[0] = (scalar *)diag; U[0] = (scalar *)upperC; Lw[0] = (scalar *)lowerC;
    for (int l = 0; l < nL; l++) {
        const gamg_level *L = &H->lev[l];
        D[l + 1] = (scalar *)malloc(sizeof(scalar) * (size_t)L->nCoarse);
        U[l + 1] = (scalar *)malloc(sizeof(scalar) * (size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1));
        Lw[l + 1] = asym ? (scalar *)malloc(sizeof(scalar) * (size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1)) : NULL;
        agglomerate_matrix(L, asym, D[l], U[l], Lw[l], D[l + 1], U[l + 1], Lw[l + 1]);
        A[l + 1] = orc_sys_create(1);
        orc_sys_set_domain(A[l + 1], 0, L->nCoarse, L->nCoarseFaces, L->cLower, L->cUpper, D[l + 1], Lw[l + 1], U[l + 1]);
    }
    /* coarsest level: dense LU (GAMGSolver.C:144-172) */
    const gamg_level *Lc = &H->lev[nL - 1];
    const int nc = Lc->nCoarse;
    scalar *dense = (scalar *)calloc((size_t)nc * (size_t)nc, sizeof(scalar));
    int *piv = (int *)malloc(sizeof(int) * (size_t)nc);
    for (int i = 0; i < nc; i++) dense[(size_t)i * nc + i] = D[nL][i];
    for (label f = 0; f < Lc->nCoarseFaces; f++) {
        dense[(size_t)Lc->cLower[f] * nc + Lc->cUpper[f]] = U[nL][f];
        dense[(size_t)Lc->cUpper[f] * nc + Lc->cLower[f]] = asym ? Lw[nL][f] : U[nL][f];
    }
    lu_factor(nc, dense, piv);

    scalar **corr = (scalar **)calloc((size_t)nL, sizeof(*corr));
    scalar **src = (scalar **)calloc((size_t)nL, sizeof(*src));
    for (int l = 0; l < nL; l++) {
        corr[l] = (scalar *)calloc((size_t)H->lev[l].nCoarse, sizeof(scalar));
        src[l] = (scalar *)calloc((size_t)H->lev[l].nCoarse, sizeof(scalar));
    }
    scalar *Apsi = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *fcorr = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *fres = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *scr1 = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *scr2 = (scalar *)calloc((size_t)n0, sizeof(scalar));

    /* ---- GAMGSolverSolve.C:70-94 ---- */
    orc_amul(A[0], psi, Apsi);
    scalar normFactor = orc_norm_factor(A[0], psi, source, Apsi, fcorr);
    perf->normFactor = normFactor;
    for (label i = 0; i < n0; i++) fres[i] = source[i] - Apsi[i];
    perf->initialResidual = orc_gSumMag(A[0], fres) / normFactor;
    perf->finalResidual = perf->initialResidual;
    if (hist && histLen > 0) hist[0] = perf->initialResidual;

    if (ctl->minIter > 0 || !conv_check(perf, ctl)) {
        const int coarsest = nL - 1;
        do {
            /* ---- Vcycle (GAMGSolverSolve.C:181-474) ---- */
            restrict_field(&H->lev[0], fres, src[0]);
            for (int l = 0; l < coarsest; l++) {
                const label nl = H->lev[l].nCoarse;
                if (ctl->nPreSweeps) {
                    memset(corr[l], 0, sizeof(scalar) * (size_t)nl);
                    orc_jacobi_smooth(A[l + 1], ctl->omega, corr[l], src[l],
                                      imin(ctl->nPreSweeps + ctl->preSweepsLevelMultiplier * l, ctl->maxPreSweeps));
                    if (doScale && l < coarsest - 1) scale_field(A[l + 1], D[l + 1], nl, corr[l], scr1, src[l]);
                    orc_amul(A[l + 1], corr[l], scr1);
                    for (label i = 0; i < nl; i++) src[l][i] -= scr1[i];
                }
                restrict_field(&H->lev[l + 1], src[l], src[l + 1]);
            }
            if (ctl->directSolveCoarsest) {
                memcpy(corr[coarsest], src[coarsest], sizeof(scalar) * (size_t)nc);
                lu_solve(nc, dense, piv, corr[coarsest]);
            } else coarsest_iterative(A[nL], asym, ctl, src[coarsest], corr[coarsest]);
            for (int l = coarsest - 1; l >= 0; l--) {
                const label nl = H->lev[l].nCoarse;
                if (ctl->nPreSweeps) memcpy(scr2, corr[l], sizeof(scalar) * (size_t)nl);
                prolong_field(&H->lev[l + 1], corr[l + 1], corr[l]);
                if (doScale && l < coarsest - 1) scale_field(A[l + 1], D[l + 1], nl, corr[l], scr1, src[l]);
                if (ctl->nPreSweeps) for (label i = 0; i < nl; i++) corr[l][i] += scr2[i];
                orc_jacobi_smooth(A[l + 1], ctl->omega, corr[l], src[l],
                                  imin(ctl->nPostSweeps + ctl->postSweepsLevelMultiplier * l, ctl->maxPostSweeps));
            }
            prolong_field(&H->lev[0], corr[0], fcorr);
            if (doScale) scale_field(A[0], D[0], n0, fcorr, Apsi, fres);
            for (label i = 0; i < n0; i++) psi[i] = psi[i] + fcorr[i];
            orc_jacobi_smooth(A[0], ctl->omega, psi, source, ctl->nFinestSweeps);
            /* ---- residual (GAMGSolverSolve.C:146-160) ---- */
            orc_amul(A[0], psi, Apsi);
            for (label i = 0; i < n0; i++) fres[i] = source[i] - Apsi[i];
            perf->finalResidual = orc_gSumMag(A[0], fres) / normFactor;
            if (hist && perf->nIterations + 1 < histLen) hist[perf->nIterations + 1] = perf->finalResidual;
        } while ((++perf->nIterations < ctl->maxIter && !conv_check(perf, ctl)) || perf->nIterations < ctl->minIter);
    }

    for (int l = 0; l <= nL; l++) { orc_sys_destroy(A[l]); if (l) { free(D[l]); free(U[l]); free(Lw[l]); } }
    for (int l = 0; l < nL; l++) { free(corr[l]); free(src[l]); }
    free(A); free(D); free(U); free(Lw); free(corr); free(src); free(dense); free(piv);
    free(Apsi); free(fcorr); free(fres); free(scr1); free(scr2);
}

/* level matrix of coarse level l+1, for unit parity tests of the engine's agglomerateMatrix kernels */
void orc_gamg_coarse_matrix(const gamg_hier *H, int upToLevel, const scalar *diag, const scalar *upperC,
                            const scalar *lowerC, scalar *cDiag, scalar *cUpper, scalar *cLower)
{
    const int asym = lowerC != NULL;
    const scalar *d = diag, *u = upperC, *lo = lowerC;
    scalar *pd = NULL, *pu = NULL, *pl = NULL;
    for (int l = 0; l <= upToLevel; l++) {
        const gamg_level *L = &H->lev[l];
        scalar *nd = (scalar *)malloc(sizeof(scalar) * (size_t)L->nCoarse);
        scalar *nu = (scalar *)malloc(sizeof(scalar) * (size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1));
        scalar *nl = asym ? (scalar *)malloc(sizeof(scalar) * (size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1)) : NULL;
        agglomerate_matrix(L, asym, d, u, lo, nd, nu, nl);
        free(pd); free(pu); free(pl);
        pd = nd; pu = nu; pl = nl; d = nd; u = nu; lo = nl;
    }
    const gamg_level *L = &H->lev[upToLevel];
    memcpy(cDiag, pd, sizeof(scalar) * (size_t)L->nCoarse);
    memcpy(cUpper, pu, sizeof(scalar) * (size_t)L->nCoarseFaces);
    if (asym && cLower) memcpy(cLower, pl, sizeof(scalar) * (size_t)L->nCoarseFaces);
    free(pd); free(pu); free(pl);
}


/* ============================================================================================
 * Coupled patches / decomposed case: every domain of an orc_system agglomerates on its own (the
 * reference runs pairGAMGAgglomeration per processor), the domains agree on when to stop
 * (GAMGAgglomeration.C:72-81), every coupled patch is agglomerated from the coarse-cell ids on
 * both of its sides (processorGAMGInterface.C:54-128, cyclicGAMGInterface.C; exchange of the ids:
 * GAMGAgglomerateLduAddressing.C:464-520), its coefficients are summed per coarse interface face
 * (GAMGInterface::agglomerateCoeffs), every level is again an orc_system with interfaces, the
 * scale factors are global sums (GAMGSolverScale.C:104-107) and the coarsest level is the global
 * system solved directly (LUscalarMatrix.C:57-150,275-400).
 * ============================================================================================ */
typedef struct {
    label nFine, nCoarse;      /* patch faces on the fine / coarse side of the level */
    label *faceRestrict;       /* [nFine] -> coarse interface face */
    label *faceCells;          /* [nCoarse] coarse cell on this side */
    /* cyclicAMI: the agglomerated AMI of the coarse side (AMIInterpolation.C:279-540), NULL otherwise */
    label *amiStart, *amiAddr; /* [nCoarse+1], [amiStart[nCoarse]] coarse face of the neighbour patch */
    scalar *amiW, *amiMagSf;   /* weights (normalised per coarse face), agglomerated face areas [nCoarse] */
} gamg_patch;

/* AMIInterpolation::agglomerate (AMIInterpolation.C:279-540, the branch without a distribution map) for one side:
 * fine faces in order, their addresses in order; an address whose coarse target is already in the coarse face's list adds
 * fineArea*weight onto it, a new one is appended; then normaliseWeights(conformal = true): every list divided by its sum
 * (AMIInterpolation.C:199-247).  Host arithmetic in the reference: a product, then an addition.                       */
static void ami_agglomerate(label nFineSrc, const label *fStart, const label *fAddr, const scalar *fW, const scalar *fMagSf,
                            const label *srcRestrict, const label *tgtRestrict, label nCoarseSrc, gamg_patch *P)
{
    label **el = (label **)calloc((size_t)(nCoarseSrc ? nCoarseSrc : 1), sizeof(label *));
    scalar **wl = (scalar **)calloc((size_t)(nCoarseSrc ? nCoarseSrc : 1), sizeof(scalar *));
    label *cnt = (label *)calloc((size_t)(nCoarseSrc ? nCoarseSrc : 1), sizeof(label));
    P->amiMagSf = (scalar *)calloc((size_t)(nCoarseSrc ? nCoarseSrc : 1), sizeof(scalar));
    for (label i = 0; i < nFineSrc; i++) P->amiMagSf[srcRestrict[i]] += fMagSf[i];
    label total = 0;
    for (label i = 0; i < nFineSrc; i++) {
        const label I = srcRestrict[i];
        const scalar fineArea = fMagSf[i];
        for (label k = fStart[i]; k < fStart[i + 1]; k++) {
            const label K = tgtRestrict[fAddr[k]];
            label j;
            for (j = 0; j < cnt[I]; j++) if (el[I][j] == K) break;
            const scalar t = fineArea * fW[k];
            if (j == cnt[I]) {
                el[I] = (label *)realloc(el[I], sizeof(label) * (size_t)(cnt[I] + 1));
                wl[I] = (scalar *)realloc(wl[I], sizeof(scalar) * (size_t)(cnt[I] + 1));
                el[I][j] = K; wl[I][j] = t; cnt[I]++; total++;
            } else wl[I][j] += t;
        }
    }
    P->amiStart = (label *)malloc(sizeof(label) * (size_t)(nCoarseSrc + 1));
    P->amiAddr = (label *)malloc(sizeof(label) * (size_t)(total ? total : 1));
    P->amiW = (scalar *)malloc(sizeof(scalar) * (size_t)(total ? total : 1));
    label at = 0;
    for (label I = 0; I < nCoarseSrc; I++) {
        P->amiStart[I] = at;
        scalar sum = 0;
        for (label j = 0; j < cnt[I]; j++) sum += wl[I][j];
        for (label j = 0; j < cnt[I]; j++) { P->amiAddr[at] = el[I][j]; P->amiW[at] = wl[I][j] / sum; at++; }
        free(el[I]); free(wl[I]);
    }
    P->amiStart[nCoarseSrc] = at;
    free(el); free(wl); free(cnt);
}

typedef struct {
    int nDomains, nLevels, forwardOut;
    gamg_level **lev;          /* [d][l] */
    gamg_patch ***patch;       /* [d][l][p] */
    int *nPatches;             /* [d] */
} gamg_sys_hier;

gamg_sys_hier *orc_gamg_build_sys_merged(const orc_system *S, const scalar *faceWeights, label nCellsInCoarsestLevel, int forwardInit, int mergeLevels);
gamg_sys_hier *orc_gamg_build_sys(const orc_system *S, const scalar *faceWeights /* per domain, concatenated */,
                                  label nCellsInCoarsestLevel, int forwardInit)
{
    return orc_gamg_build_sys_merged(S, faceWeights, nCellsInCoarsestLevel, forwardInit, 1);
}
gamg_sys_hier *orc_gamg_build_sys_merged(const orc_system *S, const scalar *faceWeights /* per domain, concatenated */,
                                         label nCellsInCoarsestLevel, int forwardInit, int mergeLevels)
{
    const int maxLevels = 50, D = S->nDomains;
    gamg_sys_hier *H = (gamg_sys_hier *)calloc(1, sizeof(*H));
    H->nDomains = D;
    H->lev = (gamg_level **)calloc((size_t)D, sizeof(*H->lev));
    H->patch = (gamg_patch ***)calloc((size_t)D, sizeof(*H->patch));
    H->nPatches = (int *)calloc((size_t)D, sizeof(int));
    scalar **w = (scalar **)calloc((size_t)D, sizeof(*w));
    const label **lo = (const label **)calloc((size_t)D, sizeof(*lo)), **up = (const label **)calloc((size_t)D, sizeof(*up));
    label *nFine = (label *)calloc((size_t)D, sizeof(label)), *nF = (label *)calloc((size_t)D, sizeof(label));
    label ***pfc = (label ***)calloc((size_t)D, sizeof(*pfc)); /* current fine faceCells [d][p] */
    label **pn = (label **)calloc((size_t)D, sizeof(*pn));     /* their sizes */
    /* cyclicAMI patches: AMI tables of the current fine level (borrowed from the system on the finest level, from the
     * previous level's gamg_patch afterwards) */
    const label ***fAs = (const label ***)calloc((size_t)D, sizeof(*fAs)), ***fAa = (const label ***)calloc((size_t)D, sizeof(*fAa));
    const scalar ***fAw = (const scalar ***)calloc((size_t)D, sizeof(*fAw)), ***fAm = (const scalar ***)calloc((size_t)D, sizeof(*fAm));
    int64_t woff = 0;
    for (int d = 0; d < D; d++) {
        const orc_domain *m = &S->dom[d];
        H->lev[d] = (gamg_level *)calloc((size_t)maxLevels, sizeof(gamg_level));
        H->patch[d] = (gamg_patch **)calloc((size_t)maxLevels, sizeof(gamg_patch *));
        H->nPatches[d] = m->nIfaces;
        w[d] = (scalar *)malloc(sizeof(scalar) * (size_t)(m->nFaces ? m->nFaces : 1));
        memcpy(w[d], faceWeights + woff, sizeof(scalar) * (size_t)m->nFaces); woff += m->nFaces;
        lo[d] = m->lower; up[d] = m->upper; nFine[d] = m->nCells; nF[d] = m->nFaces;
        pfc[d] = (label **)calloc((size_t)(m->nIfaces ? m->nIfaces : 1), sizeof(label *));
        pn[d] = (label *)calloc((size_t)(m->nIfaces ? m->nIfaces : 1), sizeof(label));
        const size_t np1 = (size_t)(m->nIfaces ? m->nIfaces : 1);
        fAs[d] = (const label **)calloc(np1, sizeof(label *)); fAa[d] = (const label **)calloc(np1, sizeof(label *));
        fAw[d] = (const scalar **)calloc(np1, sizeof(scalar *)); fAm[d] = (const scalar **)calloc(np1, sizeof(scalar *));
        for (int p = 0; p < m->nIfaces; p++) {
            pn[d][p] = m->ifaces[p].nFaces;
            pfc[d][p] = (label *)malloc(sizeof(label) * (size_t)(pn[d][p] ? pn[d][p] : 1));
            memcpy(pfc[d][p], m->ifaces[p].faceCells, sizeof(label) * (size_t)pn[d][p]);
            if (m->ifaces[p].amiStart) {
                if (!m->ifaces[p].amiMagSf || mergeLevels != 1) { fprintf(stderr, "orc_gamg_build_sys: cyclicAMI interfaces need face areas (orc_sys_set_iface_magsf) and mergeLevels 1\n"); abort(); }
                fAs[d][p] = m->ifaces[p].amiStart; fAa[d][p] = m->ifaces[p].amiAddr; fAw[d][p] = m->ifaces[p].amiW; fAm[d][p] = m->ifaces[p].amiMagSf;
            }
        }
    }
    int forward = forwardInit, nPairLevels = 0;
    while (H->nLevels < maxLevels - 1) {
        const int l = H->nLevels;
        int cont = 1;
        for (int d = 0; d < D; d++) {
            gamg_level *L = &H->lev[d][l];
            L->nFine = nFine[d]; L->nFineFaces = nF[d];
            L->restrictMap = (label *)malloc(sizeof(label) * (size_t)nFine[d]);
            L->nCoarse = pair_agglomerate(nFine[d], nF[d], lo[d], up[d], w[d], forward, L->restrictMap);
            if (!(L->nCoarse >= nCellsInCoarsestLevel) || L->nCoarse == nFine[d]) cont = 0; /* andOp over the processors */
        }
        forward = !forward;
        if (!cont) { for (int d = 0; d < D; d++) { free(H->lev[d][l].restrictMap); H->lev[d][l].restrictMap = NULL; } break; }
        for (int d = 0; d < D; d++) {
            gamg_level *L = &H->lev[d][l];
            coarse_addressing(L, lo[d], up[d]);
            scalar *cw = (scalar *)calloc((size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1), sizeof(scalar));
            for (label f = 0; f < nF[d]; f++) if (L->faceRestrict[f] >= 0) cw[L->faceRestrict[f]] += w[d][f];
            free(w[d]); w[d] = cw;
        }
        /* coupled patches: one coarse interface face per distinct (my coarse cell, neighbour's coarse cell) pair,
         * in order of first appearance (both sides visit matching faces in the same order) */
        for (int d = 0; d < D; d++) {
            const orc_domain *m = &S->dom[d];
            H->patch[d][l] = (gamg_patch *)calloc((size_t)(m->nIfaces ? m->nIfaces : 1), sizeof(gamg_patch));
            for (int p = 0; p < m->nIfaces; p++) {
                const int nd = m->ifaces[p].nbrDomain, np = m->ifaces[p].nbrPatch;
                gamg_patch *P = &H->patch[d][l][p];
                const label n = pn[d][p];
                P->nFine = n;
                P->faceRestrict = (label *)malloc(sizeof(label) * (size_t)(n ? n : 1));
                P->faceCells = (label *)malloc(sizeof(label) * (size_t)(n ? n : 1));
                label *nbr = (label *)malloc(sizeof(label) * (size_t)(n ? n : 1));
                label nc = 0;
                if (fAs[d][p]) {
                    /* cyclicAMIGAMGInterface.C:66-110: one coarse face per distinct LOCAL coarse cell, first appearance */
                    for (label i = 0; i < n; i++) {
                        const label mine = H->lev[d][l].restrictMap[pfc[d][p][i]];
                        label k;
                        for (k = 0; k < nc; k++) if (P->faceCells[k] == mine) break;
                        if (k == nc) { P->faceCells[nc] = mine; nc++; }
                        P->faceRestrict[i] = k;
                    }
                    P->nCoarse = nc;
                    free(nbr);
                    continue;
                }
                for (label i = 0; i < n; i++) {
                    const label mine = H->lev[d][l].restrictMap[pfc[d][p][i]];
                    const label theirs = H->lev[nd][l].restrictMap[pfc[nd][np][i]];
                    label k;
                    for (k = 0; k < nc; k++) if (P->faceCells[k] == mine && nbr[k] == theirs) break;
                    if (k == nc) { P->faceCells[nc] = mine; nbr[nc] = theirs; nc++; }
                    P->faceRestrict[i] = k;
                }
                P->nCoarse = nc;
                free(nbr);
            }
        }
        /* the AMI of the coarse side, from both sides' face agglomeration (cyclicAMIGAMGInterface.C:113-160 builds the
         * neighbour's the way the neighbour does; here the neighbour's own is at hand) */
        for (int d = 0; d < D; d++) {
            const orc_domain *m = &S->dom[d];
            for (int p = 0; p < m->nIfaces; p++) if (fAs[d][p]) {
                const int nd = m->ifaces[p].nbrDomain, np = m->ifaces[p].nbrPatch;
                gamg_patch *P = &H->patch[d][l][p];
                const orc_iface *me = &m->ifaces[p];
                if (me->nAmiParts > 0) {
                    /* partner side split over several domains: addresses number the pieces' faces concatenated, on the fine level
                     * and -- piece by piece, every piece with the coarse faces ITS domain built -- on the coarse level */
                    label nFineTgt = 0, nCoarseTgt = 0;
                    for (int q = 0; q < me->nAmiParts; q++) nFineTgt += pn[me->amiPartDomain[q]][me->amiPartPatch[q]];
                    label *tgtR = (label *)malloc(sizeof(label) * (size_t)(nFineTgt ? nFineTgt : 1));
                    label at = 0;
                    for (int q = 0; q < me->nAmiParts; q++) {
                        const int dq = me->amiPartDomain[q], pq = me->amiPartPatch[q];
                        const gamg_patch *T = &H->patch[dq][l][pq];
                        for (label j = 0; j < pn[dq][pq]; j++) tgtR[at++] = nCoarseTgt + T->faceRestrict[j];
                        nCoarseTgt += T->nCoarse;
                    }
                    ami_agglomerate(pn[d][p], fAs[d][p], fAa[d][p], fAw[d][p], fAm[d][p], P->faceRestrict, tgtR, P->nCoarse, P);
                    free(tgtR);
                } else
                ami_agglomerate(pn[d][p], fAs[d][p], fAa[d][p], fAw[d][p], fAm[d][p], P->faceRestrict, H->patch[nd][l][np].faceRestrict, P->nCoarse, P);
            }
        }
        for (int d = 0; d < D; d++) {
            const orc_domain *m = &S->dom[d];
            for (int p = 0; p < m->nIfaces; p++) if (fAs[d][p]) {
                const gamg_patch *P = &H->patch[d][l][p];
                fAs[d][p] = P->amiStart; fAa[d][p] = P->amiAddr; fAw[d][p] = P->amiW; fAm[d][p] = P->amiMagSf;
            }
        }
        for (int d = 0; d < D; d++) {
            const orc_domain *m = &S->dom[d];
            gamg_level *L = &H->lev[d][l];
            for (int p = 0; p < m->nIfaces; p++) {
                gamg_patch *P = &H->patch[d][l][p];
                free(pfc[d][p]);
                pfc[d][p] = (label *)malloc(sizeof(label) * (size_t)(P->nCoarse ? P->nCoarse : 1));
                memcpy(pfc[d][p], P->faceCells, sizeof(label) * (size_t)P->nCoarse);
                pn[d][p] = P->nCoarse;
            }
            nFine[d] = L->nCoarse; nF[d] = L->nCoarseFaces; lo[d] = L->cLower; up[d] = L->cUpper;
        }
        if (nPairLevels % mergeLevels) { /* combineLevels, per processor; patch maps: GAMGAgglomerateLduAddressing.C:700-760 */
            for (int d = 0; d < D; d++) {
                combine_levels(&H->lev[d][l - 1], &H->lev[d][l]);
                for (int p = 0; p < S->dom[d].nIfaces; p++) {
                    gamg_patch *P = &H->patch[d][l - 1][p], *C = &H->patch[d][l][p];
                    for (label i = 0; i < P->nFine; i++) P->faceRestrict[i] = C->faceRestrict[P->faceRestrict[i]];
                    free(P->faceCells); P->faceCells = C->faceCells; P->nCoarse = C->nCoarse;
                    free(C->faceRestrict);
                }
                free(H->patch[d][l]); H->patch[d][l] = NULL;
            }
        } else H->nLevels++;
        nPairLevels++;
    }
    H->forwardOut = forward;
    for (int d = 0; d < D; d++) {
        free(w[d]);
        for (int p = 0; p < S->dom[d].nIfaces; p++) free(pfc[d][p]);
        free(pfc[d]); free(pn[d]); free(fAs[d]); free(fAa[d]); free(fAw[d]); free(fAm[d]);
    }
    free(w); free(lo); free(up); free(nFine); free(nF); free(pfc); free(pn); free(fAs); free(fAa); free(fAw); free(fAm);
    return H;
}

int orc_gamg_sys_n_levels(const gamg_sys_hier *H) { return H->nLevels; }
void orc_gamg_sys_level_sizes(const gamg_sys_hier *H, int d, int l, label *out4)
{
    out4[0] = H->lev[d][l].nFine; out4[1] = H->lev[d][l].nFineFaces; out4[2] = H->lev[d][l].nCoarse; out4[3] = H->lev[d][l].nCoarseFaces;
}
void orc_gamg_sys_level_maps(const gamg_sys_hier *H, int d, int l, label *restrictMap, label *cLower, label *cUpper)
{
    const gamg_level *L = &H->lev[d][l];
    if (restrictMap) memcpy(restrictMap, L->restrictMap, sizeof(label) * (size_t)L->nFine);
    if (cLower) memcpy(cLower, L->cLower, sizeof(label) * (size_t)L->nCoarseFaces);
    if (cUpper) memcpy(cUpper, L->cUpper, sizeof(label) * (size_t)L->nCoarseFaces);
}
label orc_gamg_sys_patch(const gamg_sys_hier *H, int d, int l, int p, label *faceRestrict, label *faceCells)
{
    const gamg_patch *P = &H->patch[d][l][p];
    if (faceRestrict) memcpy(faceRestrict, P->faceRestrict, sizeof(label) * (size_t)P->nFine);
    if (faceCells) memcpy(faceCells, P->faceCells, sizeof(label) * (size_t)P->nCoarse);
    return P->nCoarse;
}
/* the agglomerated AMI of a cyclicAMI patch on the coarse side of level l: returns the number of addresses (-1: not an AMI
 * patch); start [nCoarse+1], addr / w [count], magSf [nCoarse] are filled when given */
label orc_gamg_sys_patch_ami(const gamg_sys_hier *H, int d, int l, int p, label *start, label *addr, scalar *w, scalar *magSf)
{
    const gamg_patch *P = &H->patch[d][l][p];
    if (!P->amiStart) return -1;
    const label na = P->amiStart[P->nCoarse];
    if (start) memcpy(start, P->amiStart, sizeof(label) * (size_t)(P->nCoarse + 1));
    if (addr) memcpy(addr, P->amiAddr, sizeof(label) * (size_t)na);
    if (w) memcpy(w, P->amiW, sizeof(scalar) * (size_t)na);
    if (magSf) memcpy(magSf, P->amiMagSf, sizeof(scalar) * (size_t)P->nCoarse);
    return na;
}
void orc_gamg_sys_free(gamg_sys_hier *H)
{
    for (int d = 0; d < H->nDomains; d++) {
        for (int l = 0; l < H->nLevels; l++) {
            gamg_level *L = &H->lev[d][l];
            free(L->restrictMap); free(L->faceRestrict); free(L->faceFlip); free(L->cLower); free(L->cUpper);
            for (int p = 0; p < H->nPatches[d]; p++) {
                gamg_patch *P = &H->patch[d][l][p];
                free(P->faceRestrict); free(P->faceCells); free(P->amiStart); free(P->amiAddr); free(P->amiW); free(P->amiMagSf);
            }
            free(H->patch[d][l]);
        }
        free(H->lev[d]); free(H->patch[d]);
    }
    free(H->lev); free(H->patch); free(H->nPatches); free(H);
}

/* level l+1 system from level l (coefficients by summation, interfaces agglomerated) */
static orc_system *coarse_system(const gamg_sys_hier *H, int l, const orc_system *F)
{
    const int D = H->nDomains;
    orc_system *C = orc_sys_create(D);
    for (int d = 0; d < D; d++) {
        const gamg_level *L = &H->lev[d][l];
        const orc_domain *fm = &F->dom[d];
        const int asym = !fm->symmetric;
        scalar *cd = (scalar *)malloc(sizeof(scalar) * (size_t)L->nCoarse);
        scalar *cu = (scalar *)malloc(sizeof(scalar) * (size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1));
        scalar *clw = asym ? (scalar *)malloc(sizeof(scalar) * (size_t)(L->nCoarseFaces ? L->nCoarseFaces : 1)) : NULL;
        agglomerate_matrix(L, asym, fm->diag, fm->upperC, asym ? fm->lowerC : NULL, cd, cu, clw);
        orc_sys_set_domain(C, d, L->nCoarse, L->nCoarseFaces, L->cLower, L->cUpper, cd, clw, cu);
        free(cd); free(cu); free(clw);
    }
    for (int d = 0; d < D; d++) {
        const orc_domain *fm = &F->dom[d];
        for (int p = 0; p < fm->nIfaces; p++) {
            const gamg_patch *P = &H->patch[d][l][p];
            scalar *cb = (scalar *)calloc((size_t)(P->nCoarse ? P->nCoarse : 1), sizeof(scalar));
            scalar *ci = (scalar *)calloc((size_t)(P->nCoarse ? P->nCoarse : 1), sizeof(scalar));
            for (label i = 0; i < P->nFine; i++) { cb[P->faceRestrict[i]] += fm->ifaces[p].bouCoeffs[i]; ci[P->faceRestrict[i]] += fm->ifaces[p].intCoeffs[i]; }
            orc_sys_add_interface(C, d, fm->ifaces[p].nbrDomain, fm->ifaces[p].nbrPatch, P->nCoarse, P->faceCells, cb, ci);
            if (P->amiStart) orc_sys_set_iface_ami(C, d, p, P->amiStart, P->amiAddr, P->amiW, NULL); /* no low-weight correction on coarse levels (AMIInterpolation.C:751) */
            orc_sys_set_iface_transform(C, d, p, fm->ifaces[p].factor);                               /* doTransform_/rank_ are the fine interface's (cyclicAMIGAMGInterfaceField.C:63-68) */
            free(cb); free(ci);
        }
    }
    for (int d = 0; d < D; d++) {      /* split partner sides: the same pieces, with the coarse interfaces' face counts */
        const orc_domain *fm = &F->dom[d];
        for (int p = 0; p < fm->nIfaces; p++)
            if (fm->ifaces[p].nAmiParts > 0) orc_sys_set_iface_ami_parts(C, d, p, fm->ifaces[p].nAmiParts, fm->ifaces[p].amiPartDomain, fm->ifaces[p].amiPartPatch);
    }
    return C;
}

static void sys_restrict(const gamg_sys_hier *H, int l, const orc_system *F, const orc_system *C, const scalar *ff, scalar *cf)
{
    for (int d = 0; d < H->nDomains; d++) restrict_field(&H->lev[d][l], ff + F->dom[d].offset, cf + C->dom[d].offset);
}
static void sys_prolong(const gamg_sys_hier *H, int l, const orc_system *F, const orc_system *C, const scalar *cf, scalar *ff)
{
    for (int d = 0; d < H->nDomains; d++) prolong_field(&H->lev[d][l], cf + C->dom[d].offset, ff + F->dom[d].offset);
}
static void sys_scale(const orc_system *A, scalar *field, scalar *Acf, const scalar *source)
{
    orc_amul(A, field, Acf);
    long double num = 0, den = 0;
    for (int64_t i = 0; i < A->nTotal; i++) { num += (long double)source[i] * field[i]; den += (long double)Acf[i] * field[i]; }
    scalar dd = (scalar)den;
    scalar sf = (scalar)num / (dd >= 0 ? dd + G_VSMALL : dd - G_VSMALL);
    for (int d = 0; d < A->nDomains; d++) {
        const orc_domain *m = &A->dom[d];
        for (label i = 0; i < m->nCells; i++) {
            const int64_t g = m->offset + i;
            field[g] = fma(sf, field[g], fma(-sf, Acf[g], source[g]) / m->diag[i]);
        }
    }
}

typedef struct { int n; scalar *dense; int *piv; } orc_lu;
orc_lu *orc_gamg_sys_coarsest_lu(const orc_system *Ac);
void orc_lu_free(orc_lu *L);

void orc_gamg_solve_sys(const gamg_sys_hier *H, const orc_system *S, scalar *psi, const scalar *source,
                        const gamg_controls *ctl, gamg_perf *perf, scalar *hist, int histLen)
{
    const int nL = H->nLevels;
    const int asym = !S->dom[0].symmetric;
    const int doScale = ctl->scaleCorrection < 0 ? !asym : ctl->scaleCorrection;
    memset(perf, 0, sizeof(*perf));
    if (nL < 1) { perf->singular = 1; return; }
    const orc_system **A = (const orc_system **)calloc((size_t)nL + 1, sizeof(*A));
    A[0] = S;
    for (int l = 0; l < nL; l++) A[l + 1] = coarse_system(H, l, A[l]);
    /* coarsest level: the global matrix, dense LU */
    const orc_system *Ac = A[nL];
    const int nc = (int)Ac->nTotal;
    /* (only when it is used: with directSolveCoarsest off the coarsest level is solved by ICCG / BICCG below, and a system with
     *  cyclicAMI interfaces has no such dense form in the reference -- LUscalarMatrix.C:246-251 casts every interface to a cyclic one) */
    orc_lu *LU = ctl->directSolveCoarsest ? orc_gamg_sys_coarsest_lu(Ac) : NULL;
    scalar *dense = LU ? LU->dense : NULL; int *piv = LU ? LU->piv : NULL;

    const int64_t n0 = S->nTotal;
    scalar **corr = (scalar **)calloc((size_t)nL, sizeof(*corr));
    scalar **src = (scalar **)calloc((size_t)nL, sizeof(*src));
    for (int l = 0; l < nL; l++) {
        corr[l] = (scalar *)calloc((size_t)A[l + 1]->nTotal, sizeof(scalar));
        src[l] = (scalar *)calloc((size_t)A[l + 1]->nTotal, sizeof(scalar));
    }
    scalar *Apsi = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *fcorr = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *fres = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *scr1 = (scalar *)calloc((size_t)n0, sizeof(scalar));
    scalar *scr2 = (scalar *)calloc((size_t)n0, sizeof(scalar));

    orc_amul(S, psi, Apsi);
    scalar normFactor = orc_norm_factor(S, psi, source, Apsi, fcorr);
    perf->normFactor = normFactor;
    for (int64_t i = 0; i < n0; i++) fres[i] = source[i] - Apsi[i];
    perf->initialResidual = orc_gSumMag(S, fres) / normFactor;
    perf->finalResidual = perf->initialResidual;
    if (hist && histLen > 0) hist[0] = perf->initialResidual;

    if (ctl->minIter > 0 || !conv_check(perf, ctl)) {
        const int coarsest = nL - 1;
        do {
            sys_restrict(H, 0, A[0], A[1], fres, src[0]);
            for (int l = 0; l < coarsest; l++) {
                const int64_t nl = A[l + 1]->nTotal;
                if (ctl->nPreSweeps) {
                    memset(corr[l], 0, sizeof(scalar) * (size_t)nl);
                    orc_jacobi_smooth(A[l + 1], ctl->omega, corr[l], src[l],
                                      imin(ctl->nPreSweeps + ctl->preSweepsLevelMultiplier * l, ctl->maxPreSweeps));
                    if (doScale && l < coarsest - 1) sys_scale(A[l + 1], corr[l], scr1, src[l]);
                    orc_amul(A[l + 1], corr[l], scr1);
                    for (int64_t i = 0; i < nl; i++) src[l][i] -= scr1[i];
                }
                sys_restrict(H, l + 1, A[l + 1], A[l + 2], src[l], src[l + 1]);
            }
            if (ctl->directSolveCoarsest) {
                memcpy(corr[coarsest], src[coarsest], sizeof(scalar) * (size_t)nc);
                lu_solve(nc, dense, piv, corr[coarsest]);
            } else coarsest_iterative(A[nL], asym, ctl, src[coarsest], corr[coarsest]);
            for (int l = coarsest - 1; l >= 0; l--) {
                const int64_t nl = A[l + 1]->nTotal;
                if (ctl->nPreSweeps) memcpy(scr2, corr[l], sizeof(scalar) * (size_t)nl);
                sys_prolong(H, l + 1, A[l + 1], A[l + 2], corr[l + 1], corr[l]);
                if (doScale && l < coarsest - 1) sys_scale(A[l + 1], corr[l], scr1, src[l]);
                if (ctl->nPreSweeps) for (int64_t i = 0; i < nl; i++) corr[l][i] += scr2[i];
                orc_jacobi_smooth(A[l + 1], ctl->omega, corr[l], src[l],
                                  imin(ctl->nPostSweeps + ctl->postSweepsLevelMultiplier * l, ctl->maxPostSweeps));
            }
            sys_prolong(H, 0, A[0], A[1], corr[0], fcorr);
            if (doScale) sys_scale(A[0], fcorr, Apsi, fres);
            for (int64_t i = 0; i < n0; i++) psi[i] = psi[i] + fcorr[i];
            orc_jacobi_smooth(S, ctl->omega, psi, source, ctl->nFinestSweeps);
            orc_amul(S, psi, Apsi);
            for (int64_t i = 0; i < n0; i++) fres[i] = source[i] - Apsi[i];
            perf->finalResidual = orc_gSumMag(S, fres) / normFactor;
            if (hist && perf->nIterations + 1 < histLen) hist[perf->nIterations + 1] = perf->finalResidual;
        } while ((++perf->nIterations < ctl->maxIter && !conv_check(perf, ctl)) || perf->nIterations < ctl->minIter);
    }
    for (int l = 1; l <= nL; l++) orc_sys_destroy((orc_system *)A[l]);
    for (int l = 0; l < nL; l++) { free(corr[l]); free(src[l]); }
    free(A); free(corr); free(src); orc_lu_free(LU);
    free(Apsi); free(fcorr); free(fres); free(scr1); free(scr2);
}

/* ---- primitives of the multi-domain GAMG exposed one by one: oracle/ref_shim drives them from the REFERENCE's own
 *      GAMGSolverSolve.C (solve / Vcycle / initVcycle / solveCoarsestLevel compiled from /root/reference) ------------- */
orc_system *orc_gamg_sys_coarse_system(const gamg_sys_hier *H, int l, const orc_system *F) { return coarse_system(H, l, F); }
void orc_gamg_sys_restrict(const gamg_sys_hier *H, int l, const orc_system *F, const orc_system *C, const scalar *ff, scalar *cf) { sys_restrict(H, l, F, C, ff, cf); }
void orc_gamg_sys_prolong(const gamg_sys_hier *H, int l, const orc_system *F, const orc_system *C, const scalar *cf, scalar *ff) { sys_prolong(H, l, F, C, cf, ff); }
void orc_gamg_sys_scale(const orc_system *A, scalar *field, scalar *Acf, const scalar *source) { sys_scale(A, field, Acf, source); }
orc_lu *orc_gamg_sys_coarsest_lu(const orc_system *Ac)
{
    orc_lu *L = (orc_lu *)calloc(1, sizeof(*L));
    const int nc = (int)Ac->nTotal;
    L->n = nc;
    L->dense = (scalar *)calloc((size_t)nc * (size_t)nc, sizeof(scalar));
    L->piv = (int *)malloc(sizeof(int) * (size_t)(nc ? nc : 1));
    for (int d = 0; d < Ac->nDomains; d++) {
        const orc_domain *m = &Ac->dom[d];
        const int64_t o = m->offset;
        for (label i = 0; i < m->nCells; i++) L->dense[(size_t)(o + i) * nc + (size_t)(o + i)] = m->diag[i];
        for (label f = 0; f < m->nFaces; f++) {
            L->dense[(size_t)(o + m->lower[f]) * nc + (size_t)(o + m->upper[f])] = m->upperC[f];
            L->dense[(size_t)(o + m->upper[f]) * nc + (size_t)(o + m->lower[f])] = m->lowerC[f];
        }
        for (int p = 0; p < m->nIfaces; p++) {
            const orc_iface *me = &m->ifaces[p];
            const orc_domain *nb = &Ac->dom[me->nbrDomain];
            const orc_iface *ot = &nb->ifaces[me->nbrPatch];
            if (me->amiStart) { fprintf(stderr, "orc_gamg_sys_coarsest_lu: cyclicAMI interfaces have no direct coarsest solve (LUscalarMatrix.C:246-251): directSolveCoarsest 0\n"); abort(); }
            for (label k = 0; k < me->nFaces; k++)
                L->dense[(size_t)(o + me->faceCells[k]) * nc + (size_t)(nb->offset + ot->faceCells[k])] -= me->bouCoeffs[k];
        }
    }
    lu_factor(nc, L->dense, L->piv);
    return L;
}
void orc_lu_solve(const orc_lu *L, scalar *b) { lu_solve(L->n, L->dense, L->piv, b); }
void orc_lu_free(orc_lu *L) { if (!L) return; free(L->dense); free(L->piv); free(L); }
