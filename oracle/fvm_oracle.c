/*
 * fvm_oracle.c -- CPU ORACLE for the fvMatrix assembly sweeps (test infrastructure, NOT product
 * code; PARITY UNPINNED, see ldu_oracle.c).  Scalar fields.  Paths relative to
 * /root/reference/src/ .
 *   negSumDiag / sumDiag / sumMagOffDiag  OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixOperations.C:36-106
 *                                          (row order of lduAddressingFunctors.H:10-64: own faces, then losort)
 *   fvm::laplacian (uncorrected)          finiteVolume/finiteVolume/laplacianSchemes/gaussLaplacianScheme/gaussLaplacianScheme.C:44-88
 *   fvm::div                              finiteVolume/finiteVolume/convectionSchemes/gaussConvectionScheme/gaussConvectionScheme.C:74-115
 *   addToInternalField (boundary diag /   finiteVolume/fvMatrices/fvMatrix/fvMatrix.C:38-124,208-226,289-349
 *     source), per unique patch cell in ascending patch-face order
 *   relax                                 finiteVolume/fvMatrices/fvMatrix/fvMatrix.C:1087-1345 (+ functors :978-1085)
 *   surfaceIntegrate                      finiteVolume/finiteVolume/fvc/fvcSurfaceIntegrate.C:40-96
 *   face interpolation                    finiteVolume/interpolation/surfaceInterpolation/surfaceInterpolationScheme/surfaceInterpolationScheme.C:274-352
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int32_t label;
typedef double scalar;

void orc_addr_tables(label nCells, label nFaces, const label *lower, const label *upper, label *losort,
                     label *ownerStart, label *losortStart);

typedef struct { label *losort, *os, *ls; } tables;
static tables mk(label n, label nf, const label *lo, const label *up)
{
    tables t;
    t.losort = (label *)malloc(sizeof(label) * (size_t)(nf ? nf : 1));
    t.os = (label *)malloc(sizeof(label) * (size_t)(n + 1));
    t.ls = (label *)malloc(sizeof(label) * (size_t)(n + 1));
    orc_addr_tables(n, nf, lo, up, t.losort, t.os, t.ls);
    return t;
}
static void rel(tables t) { free(t.losort); free(t.os); free(t.ls); }

/* kind: 0 sumDiag (diag += lower(own) + upper(nei)), 1 negSumDiag, 2 sumMagOffDiag (out += |upper|(own) + |lower|(nei)) */
void orc_row_face_op(int kind, label n, label nf, const label *lo, const label *up, const scalar *lowerC,
                     const scalar *upperC, scalar *inout)
{
    tables t = mk(n, nf, lo, up);
    for (label c = 0; c < n; c++) {
        scalar out = inout[c];
        for (label j = t.os[c]; j < t.os[c + 1]; j++)
            out += kind == 0 ? lowerC[j] : kind == 1 ? -lowerC[j] : fabs(upperC[j]);
        for (label j = t.ls[c]; j < t.ls[c + 1]; j++) {
            label g = t.losort[j];
            out += kind == 0 ? upperC[g] : kind == 1 ? -upperC[g] : fabs(lowerC[g]);
        }
        inout[c] = out;
    }
    rel(t);
}

void orc_fvm_laplacian(label n, label nf, const label *lo, const label *up, const scalar *deltaCoeffs,
                       const scalar *gammaMagSf, scalar *upperOut, scalar *diagOut)
{
    for (label f = 0; f < nf; f++) upperOut[f] = deltaCoeffs[f] * gammaMagSf[f];
    for (label c = 0; c < n; c++) diagOut[c] = 0;
    orc_row_face_op(1, n, nf, lo, up, upperOut, upperOut, diagOut); /* symmetric: lower aliases upper */
}

void orc_fvm_div(label n, label nf, const label *lo, const label *up, const scalar *weights, const scalar *faceFlux,
                 scalar *lowerOut, scalar *upperOut, scalar *diagOut)
{
    for (label f = 0; f < nf; f++) { lowerOut[f] = -weights[f] * faceFlux[f]; upperOut[f] = lowerOut[f] + faceFlux[f]; }
    for (label c = 0; c < n; c++) diagOut[c] = 0;
    orc_row_face_op(1, n, nf, lo, up, lowerOut, upperOut, diagOut);
}

/* intf[cell] (+|-)= sum of f(pf[face]) over the patch faces of the cell, ascending patch-face index.
 * fn: 0 identity, 1 negate, 2 magnitude                                                             */
void orc_patch_add(label nPatchFaces, const label *faceCells, const scalar *pf, int fn, label nCells, scalar *intf)
{
    /* group by cell with a stable sort == patchSortAddr; sequential accumulation per cell */
    (void)nCells;
    for (label i = 0; i < nPatchFaces; i++) {
        scalar v = pf[i];
        intf[faceCells[i]] += fn == 0 ? v : fn == 1 ? -v : fabs(v);
    }
}

/* relax, scalar Type.  coupled[p] != 0: processor-like patch. */
void orc_relax(label n, label nf, const label *lo, const label *up, scalar alpha, scalar *diag,
               const scalar *lowerC, const scalar *upperC, scalar *source, const scalar *psi, int nPatches,
               const label *patchSizes, const label *const *faceCells, const scalar *const *iCoeffs,
               const scalar *const *bCoeffs, const int *coupled)
{
    if (alpha <= 0) return;
    scalar *D0 = (scalar *)malloc(sizeof(scalar) * (size_t)n);
    scalar *sumOff = (scalar *)calloc((size_t)n, sizeof(scalar));
    memcpy(D0, diag, sizeof(scalar) * (size_t)n);
    orc_row_face_op(2, n, nf, lo, up, lowerC, upperC, sumOff);
    for (int p = 0; p < nPatches; p++) {
        if (!patchSizes[p]) continue;
        if (coupled[p]) {
            orc_patch_add(patchSizes[p], faceCells[p], iCoeffs[p], 0, n, diag);
            orc_patch_add(patchSizes[p], faceCells[p], bCoeffs[p], 2, n, sumOff);
        } else {
            orc_patch_add(patchSizes[p], faceCells[p], iCoeffs[p], 2, n, diag); /* cmptMax(cmptMag) */
        }
    }
    for (label c = 0; c < n; c++) { scalar d = fabs(diag[c]); diag[c] = (d > sumOff[c] ? d : sumOff[c]); }
    for (label c = 0; c < n; c++) diag[c] /= alpha;
    for (int p = 0; p < nPatches; p++) {
        if (!patchSizes[p]) continue;
        orc_patch_add(patchSizes[p], faceCells[p], iCoeffs[p], 1, n, diag); /* -component0 | -cmptMin */
    }
    for (label c = 0; c < n; c++) source[c] = fma(diag[c] - D0[c], psi[c], source[c]);
    free(D0); free(sumOff);
}

/* ivf[c] = sum_own ssf[f] - sum_nei ssf[losort] (divided by V if vol != NULL: fvc::surfaceIntegrate) */
void orc_surface_integrate(label n, label nf, const label *lo, const label *up, const scalar *ssf, const scalar *vol, scalar *ivf)
{
    tables t = mk(n, nf, lo, up);
    for (label c = 0; c < n; c++) {
        scalar out = 0;
        for (label j = t.os[c]; j < t.os[c + 1]; j++) out += ssf[j];
        for (label j = t.ls[c]; j < t.ls[c + 1]; j++) out -= ssf[t.losort[j]];
        ivf[c] = vol ? out / vol[c] : out;
    }
    rel(t);
}

/* sf[f] = lambda[f]*(phi[P] - phi[N]) + phi[N] */
void orc_face_interpolate(label nf, const label *lo, const label *up, const scalar *lambda, const scalar *phi, scalar *sf)
{
    for (label f = 0; f < nf; f++) sf[f] = fma(lambda[f], phi[lo[f]] - phi[up[f]], phi[up[f]]);
}
