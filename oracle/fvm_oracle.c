/*
 * fvm_oracle.c -- CPU ORACLE for the fvMatrix assembly sweeps (test infrastructure, NOT product code: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it).  Scalar fields.  Paths relative to /root/reference/src/ .
 * PINNED by the reference's own functors: every loop below whose reference form is a device functor (patch add / boundary source /
 * relax functors of fvMatrix.C, surfaceIntegrate, the face interpolate, gaussGrad, faceH, the limitedLinear limiter and weights,
 * setValues) is reproduced BIT FOR BIT against those functors compiled from /root/reference on the reference's own Vector / Scalar
 * primitives (oracle/ref_shim/ref_fvm_tu.cpp -> oracle/_ref/libref_fvm.so; frozen in tests/golden/golden_ref_fvm.npz;
 * tests/test_oracle.py).  Rounding rule the compiled functors show (LLVM -- the code generator family of the reference's nvcc; g++ agrees on
 * all but the limitedLinear weights functor, see oracle/Makefile): inside ONE functor a*b + c is one fma,
 * and of a sum of two products the FIRST is fused and the second rounded -- a*b + c*d = fma(a, b, c*d), a*b - c*d = fma(a, b, -(c*d)),
 * x*x' + y*y' + z*z' = fma(z, z', fma(x, x', y*y')); operations the reference spells as separate FIELD operations (gpuField
 * operators = one thrust::transform each, fields/Fields/gpuField/gpuFieldFunctionsM.C:283-340) round separately.
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

/* coupled part of fvMatrix::addBoundarySource (fvMatrix.C:245-286,318-346): fvMatrixAddBoundarySourceFunctor's
 * out += cmptMultiply(pbc[face], pnf[face]) per patch face in ascending order -- one fma inside the functor (pinned) */
void orc_patch_add_product(label nPatchFaces, const label *faceCells, const scalar *pf, const scalar *q, int fn, scalar *intf)
{
    for (label i = 0; i < nPatchFaces; i++) {
        intf[faceCells[i]] = fn == 0 ? fma(pf[i], q[i], intf[faceCells[i]]) : fma(-pf[i], q[i], intf[faceCells[i]]);
    }
}

/* boundary part of fvMatrix::flux (fvMatrix.C:1621-1653): InternalContrib = internalCoeffs*patchInternalField,
 * NeighbourContrib = boundaryCoeffs (*patchNeighbourField when coupled), flux = InternalContrib - NeighbourContrib.
 * Three separate field operations in the reference, hence three roundings (no fma).                                */
void orc_patch_flux(label nPatchFaces, const label *faceCells, const scalar *ic, const scalar *bc, const scalar *psi,
                    const scalar *psiNbr, scalar *out)
{
    label i;
    for (i = 0; i < nPatchFaces; i++) {
        volatile scalar inContrib = ic[i] * psi[faceCells[i]];
        volatile scalar nbContrib = psiNbr ? bc[i] * psiNbr[i] : bc[i];
        out[i] = inContrib - nbContrib;
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
    /* S += (D - D0)*psi_.internalField()  (fvMatrix.C:1344): three field operations, three roundings */
    for (label c = 0; c < n; c++) { const scalar dd = diag[c] - D0[c]; const scalar t = dd * psi[c]; source[c] = source[c] + t; }
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

/* ---- fvm::ddt, Euler (finiteVolume/ddtSchemes/EulerDdtScheme/EulerDdtScheme.C fvmDdt(vf) / fvmDdt(rho, vf)):
 *   diag = rDeltaT*rho*V ;  source = rDeltaT*rho*psi0*V   (left-to-right products, as written there)            */
void orc_fvm_ddt_euler(label n, scalar rDeltaT, scalar rho, const scalar *vol, const scalar *psiOld, scalar *diag, scalar *source)
{
    for (label c = 0; c < n; c++) {
        diag[c] = (rDeltaT * rho) * vol[c];
        source[c] = ((rDeltaT * rho) * psiOld[c]) * vol[c];
    }
}

/* ---- fvm::ddt(rho, vf), Euler, with a density FIELD (EulerDdtScheme.C:403-440: rhoPimpleFoam's fvm::ddt(rho, U), fvm::ddt(rho, he),
 *      and fvm::ddt(psi, p) with psi in rho's place):
 *   fvm.diag()   = rDeltaT*rho.internalField()*mesh().Vsc()
 *   fvm.source() = rDeltaT*rho.oldTime().internalField()*vf.oldTime().internalField()*mesh().Vsc()
 * every product a field operation, left to right                                                                    */
void orc_fvm_ddt_euler_rho(label n, scalar rDeltaT, const scalar *rho, const scalar *rhoOld, const scalar *vol, const scalar *psiOld,
                           scalar *diag, scalar *source)
{
    for (label c = 0; c < n; c++) {
        diag[c] = (rDeltaT * rho[c]) * vol[c];
        source[c] = ((rDeltaT * rhoOld[c]) * psiOld[c]) * vol[c];
    }
}

/* ---- implicit / explicit source terms (finiteVolume/fvm/fvmSup.C):
 *   fvm::Su   (:34-54)    source -= V*su
 *   fvm::Sp   (:100-122)  diag   += V*sp            (field sp; :150-170 the same with a dimensionedScalar: sp == NULL, spValue)
 *   fvm::SuSp (:190-214)  diag   += V*max(susp, 0) ; source -= V*min(susp, 0)*vf    (products left to right)          */
void orc_fvm_su(label n, const scalar *vol, const scalar *su, scalar *source)
{
    for (label c = 0; c < n; c++) source[c] -= vol[c] * su[c];
}
void orc_fvm_sp(label n, const scalar *vol, const scalar *sp, scalar spValue, scalar *diag)
{
    for (label c = 0; c < n; c++) diag[c] += vol[c] * (sp ? sp[c] : spValue);
}
void orc_fvm_susp(label n, const scalar *vol, const scalar *susp, const scalar *vf, scalar *diag, scalar *source)
{
    for (label c = 0; c < n; c++) {
        const scalar mx = susp[c] > 0 ? susp[c] : 0.0, mn = susp[c] < 0 ? susp[c] : 0.0;   /* max(susp, 0), min(susp, 0) */
        diag[c] += vol[c] * mx;
        source[c] -= (vol[c] * mn) * vf[c];
    }
}

/* ---- face flux of an interpolated cell vector, internal faces:  phi[f] = Sf[f] & linear.interpolate(V)[f]  with V = U or
 *      V = rho*U (a cell field: the product is rounded per cell, as rhoU0 = rho.oldTime()*U.oldTime() is, EulerDdtScheme.C:683-686);
 *      the interpolate one fma per component (surfaceInterpolationScheme.C:275-280), a & b = ax*bx + ay*by + az*bz = fma(az, bz, fma(ax, bx, ay*by))
 *      (the first of two products fused, as the compiled reference shows); optionally  + addA[f]*addB[f]  (a rounded field product, e.g. rhorAUf*fvc::ddtCorr(rho, U, phi), pEqn.H:49-56) or
 *      + addA[f]                                                                                                           */
static scalar flux_face(label f, const label *lo, const label *up, const scalar *lambda, const scalar *sx, const scalar *sy, const scalar *sz,
                        const scalar *vx, const scalar *vy, const scalar *vz, const scalar *sc, const scalar *addA, const scalar *addB)
{
    const label P = lo[f], N = up[f];
    scalar px = vx[P], py = vy[P], pz = vz[P], nx = vx[N], ny = vy[N], nz = vz[N];
    if (sc) { px = sc[P] * px; py = sc[P] * py; pz = sc[P] * pz; nx = sc[N] * nx; ny = sc[N] * ny; nz = sc[N] * nz; }
    const scalar ix = fma(lambda[f], px - nx, nx), iy = fma(lambda[f], py - ny, ny), iz = fma(lambda[f], pz - nz, nz);
    scalar d = fma(iz, sz[f], fma(ix, sx[f], iy * sy[f]));   /* Vector operator& (VectorI.H:129-132) as compiled: pinned */
    if (addA) d = d + (addB ? addA[f] * addB[f] : addA[f]);
    return d;
}
/* phi (face) and, when div != NULL, fvc::surfaceIntegrate(phi) in the reference's row order (own faces ascending +, then the
 * losort faces -, fvcSurfaceIntegrate.C:40-96), divided by V when vol != NULL -- phiHbyA and fvc::div(phiHbyA) of pEqn.H:49-71 */
void orc_flux_div(label n, label nf, const label *lo, const label *up, const scalar *lambda, const scalar *sx, const scalar *sy, const scalar *sz,
                  const scalar *vx, const scalar *vy, const scalar *vz, const scalar *sc, const scalar *addA, const scalar *addB,
                  scalar *phi, const scalar *vol, scalar *div)
{
    for (label f = 0; f < nf; f++) phi[f] = flux_face(f, lo, up, lambda, sx, sy, sz, vx, vy, vz, sc, addA, addB);
    if (div) orc_surface_integrate(n, nf, lo, up, phi, vol, div);
}

/* ---- fvc::ddtCorr(rho, U, phi), Euler (EulerDdtScheme.C:663-720, first branch: U a velocity, phi a mass flux; rho == NULL: the
 *      incompressible fvcDdtPhiCorr(U, phi) :523-551), internal faces:
 *   phiCorr = phi.oldTime() - (mesh().Sf() & fvc::interpolate(rho.oldTime()*U.oldTime()))
 *   result  = fvcDdtPhiCoeff(rhoU0, phi.oldTime(), phiCorr)*rDeltaT*phiCorr
 *   fvcDdtPhiCoeff = 1 - min(mag(phiCorr)/(mag(phi) + SMALL), 1)                       (ddtScheme.C:139-174; SMALL = 1e-15)   */
void orc_ddt_phi_corr(label nf, const label *lo, const label *up, scalar rDeltaT, const scalar *lambda, const scalar *sx, const scalar *sy,
                      const scalar *sz, const scalar *ux, const scalar *uy, const scalar *uz, const scalar *rhoOld, const scalar *phiOld,
                      scalar *out)
{
    for (label f = 0; f < nf; f++) {
        const scalar phiCorr = phiOld[f] - flux_face(f, lo, up, lambda, sx, sy, sz, ux, uy, uz, rhoOld, NULL, NULL);
        const scalar q = fabs(phiCorr) / (fabs(phiOld[f]) + 1e-15);
        const scalar coeff = 1.0 - (q < 1.0 ? q : 1.0);
        out[f] = (coeff * rDeltaT) * phiCorr;
    }
}

/* ---- upwind weights: pos(faceFlux) (interpolation/surfaceInterpolation/limitedSchemes/upwind/upwind.H:limiter 0 =>
 *      weights = pos(flux), limitedSurfaceInterpolationScheme.C:177-187)                                          */
void orc_upwind_weights(label nf, const scalar *faceFlux, scalar *w)
{
    for (label f = 0; f < nf; f++) w[f] = faceFlux[f] >= 0 ? 1.0 : 0.0;
}

/* ---- limitedLinear(k) weights for a scalar field:
 *   r        LimitedScheme/NVDTVD.H r(): gradf = phiN-phiP; gradcf = d & (flux>0 ? gradcP : gradcN), d = C[N]-C[P];
 *            |gradcf| >= 1000|gradf| ? 2*1000*sign(gradcf)*sign(gradf)-1 : 2*(gradcf/gradf)-1
 *   limiter  limitedLinear/limitedLinear.H:79-97  max(min(twoByk*r, 1), 0), twoByk = 2/max(k, SMALL)
 *   weights  limitedSurfaceInterpolationScheme.C:177-187  lim*CDweight + (1-lim)*pos(flux)
 * Products followed by an add are fused like nvcc contracts them (a*b + c -> fma).                                */
static scalar sgn(scalar x) { return x >= 0 ? 1.0 : -1.0; }  /* Scalar.H sign() */
void orc_limited_linear_weights(label nf, const label *lo, const label *up, scalar k, const scalar *cdWeights,
                                const scalar *faceFlux, const scalar *phi, const scalar *gx, const scalar *gy,
                                const scalar *gz, const scalar *Cx, const scalar *Cy, const scalar *Cz,
                                scalar *w, scalar *limiterOut)
{
    const scalar twoByk = 2.0 / (k > 1e-15 ? k : 1e-15); /* SMALL = 1e-15 (doubleScalarSMALL) */
    for (label f = 0; f < nf; f++) {
        const label P = lo[f], N = up[f];
        const scalar gradf = phi[N] - phi[P];
        const scalar dx = Cx[N] - Cx[P], dy = Cy[N] - Cy[P], dz = Cz[N] - Cz[P];
        const label c = faceFlux[f] > 0 ? P : N;
        const scalar gradcf = fma(dz, gz[c], fma(dx, gx[c], dy * gy[c]));
        scalar r;
        if (fabs(gradcf) >= 1000 * fabs(gradf)) r = 2 * 1000 * sgn(gradcf) * sgn(gradf) - 1;
        else r = fma(2.0, gradcf / gradf, -1.0);
        scalar lim = twoByk * r;
        lim = lim < 1 ? lim : 1;
        lim = lim > 0 ? lim : 0;
        if (limiterOut) limiterOut[f] = lim;
        const scalar pos = faceFlux[f] >= 0 ? 1.0 : 0.0;
        w[f] = fma(lim, cdWeights[f], (1.0 - lim) * pos);
    }
}

/* ---- fvc::grad, Gauss (finiteVolume/gradSchemes/gaussGrad/gaussGrad.C:27-90 functor, :143-250 gradf):
 *   grad[c] = ( sum_{own faces, ascending} Sf*ssf - sum_{neighbour faces, losort order} Sf*ssf ) / V
 * internal-face part; the boundary faces are added per patch with orc_patch_add before the division when present
 * (pass vol = NULL here, add, then divide).  Component arrays (Sfx, Sfy, Sfz; gx, gy, gz).                       */
void orc_gauss_grad(label n, label nf, const label *lo, const label *up, const scalar *Sfx, const scalar *Sfy,
                    const scalar *Sfz, const scalar *ssf, const scalar *vol, scalar *gx, scalar *gy, scalar *gz)
{
    tables t = mk(n, nf, lo, up);
    for (label c = 0; c < n; c++) {
        scalar ox = 0, oy = 0, oz = 0;
        for (label j = t.os[c]; j < t.os[c + 1]; j++) { ox = fma(Sfx[j], ssf[j], ox); oy = fma(Sfy[j], ssf[j], oy); oz = fma(Sfz[j], ssf[j], oz); }
        for (label j = t.ls[c]; j < t.ls[c + 1]; j++) {
            const label f = t.losort[j];
            ox = fma(-Sfx[f], ssf[f], ox); oy = fma(-Sfy[f], ssf[f], oy); oz = fma(-Sfz[f], ssf[f], oz);
        }
        gx[c] = vol ? ox / vol[c] : ox; gy[c] = vol ? oy / vol[c] : oy; gz[c] = vol ? oz / vol[c] : oz;
    }
    rel(t);
}

/* out = a*x + b*y : fvMatrix::operator+=, -=, *= on the coefficient arrays (fvMatrix.C:700-1000) */
void orc_axpby(label n, scalar a, const scalar *x, scalar b, const scalar *y, scalar *out)
{
    for (label i = 0; i < n; i++) out[i] = fma(a, x[i], b * y[i]);
}

/* ---- non-orthogonal correction of fvm::laplacian (gaussLaplacianSchemes.C:64-90, correctedSnGrad.C:45-65):
 *   flux[f] = gammaMagSf[f] * ( corrVec[f] & (lambda[f]*(grad[P] - grad[N]) + grad[N]) )        internal faces
 * the interpolate is one fma per component (surfaceInterpolationScheme.C:275-280), the dot product ax*bx + ay*by + az*bz
 * = fma(az, bz, fma(ax, bx, ay*by)), the product with gammaMagSf a separate (rounded) field operation.                       */
void orc_sngrad_correction_flux(label nf, const label *lo, const label *up, const scalar *cvx, const scalar *cvy, const scalar *cvz,
                                const scalar *lambda, const scalar *gx, const scalar *gy, const scalar *gz, const scalar *gammaMagSf,
                                scalar *flux)
{
    for (label f = 0; f < nf; f++) {
        const label P = lo[f], N = up[f];
        const scalar fx = fma(lambda[f], gx[P] - gx[N], gx[N]), fy = fma(lambda[f], gy[P] - gy[N], gy[N]), fz = fma(lambda[f], gz[P] - gz[N], gz[N]);
        const scalar corr = fma(cvz[f], fz, fma(cvx[f], fx, cvy[f] * fy));
        flux[f] = gammaMagSf ? gammaMagSf[f] * corr : corr;
    }
}
/* the same on a coupled patch: pLambda*patchInternalField + (1 - pLambda)*patchNeighbourField as separate field operations
 * (surfaceInterpolationScheme.C:360-365)                                                                            */
void orc_patch_sngrad_correction_flux(label n, const label *faceCells, const scalar *cvx, const scalar *cvy, const scalar *cvz,
                                      const scalar *w, const scalar *gx, const scalar *gy, const scalar *gz, const scalar *nx,
                                      const scalar *ny, const scalar *nz, const scalar *gammaMagSf, scalar *flux)
{
    for (label i = 0; i < n; i++) {
        const label c = faceCells[i];
        const scalar m = 1.0 - w[i];
        const scalar ax = w[i] * gx[c], ay = w[i] * gy[c], az = w[i] * gz[c];
        const scalar bx = m * nx[i], by = m * ny[i], bz = m * nz[i];
        const scalar fx = ax + bx, fy = ay + by, fz = az + bz;
        const scalar corr = fma(cvz[i], fz, fma(cvx[i], fx, cvy[i] * fy));
        flux[i] = gammaMagSf ? gammaMagSf[i] * corr : corr;
    }
}
/* inout -= x*y (product rounded first) */
void orc_submul(label n, const scalar *x, const scalar *y, scalar *io)
{
    for (label i = 0; i < n; i++) { const scalar t = x[i] * y[i]; io[i] -= t; }
}

/* ---- fvMatrix::setReference (fvMatrices/fvMatrix/fvMatrix.C:964-981): source[celli] += diag[celli]*value; diag[celli] *= 2
 *      -- two host-side get/set pairs in the reference (source().set(celli, source().get(celli) + diag().get(celli)*value)): HOST
 *      code of the reference, built for baseline x86-64 (wmake/rules/linux64Nvcc/c++: -m64, no -march), which has no fma
 *      instruction: the product is rounded, then added                                                                         */
void orc_set_reference(label celli, scalar value, scalar *diag, scalar *source)
{
    if (celli < 0) return;
    { const scalar t = diag[celli] * value; source[celli] = source[celli] + t; }
    diag[celli] = 2 * diag[celli];
}

/* ---- fvMatrix::setValues (fvMatrix.C:454-656, setValuesFromList; functors :352-452 pinned by oracle/_ref/libref_fvm.so):
 *   psi[cellLabels[i]] = values[i];  source[cellLabels[i]] = values[i]*diag[cellLabels[i]]
 *   every row that is NOT set:  source -= lower[face]*value[nei]   over its own faces whose neighbour is set (ascending),
 *                               source -= upper[face]*value[own]   over its neighbour-side faces (losort order) whose owner is set
 *                               (each one fma inside the functor)
 *   upper[face] = 0 where the OWNER is set;  lower[face] = 0 where the NEIGHBOUR is set -- the non-const lower() of the reference
 *   makes a symmetric matrix asymmetric here, so lower is always an output (lowerIn NULL: a copy of upperIn is the input);
 *   internalCoeffs / boundaryCoeffs of every patch face whose cell is set = 0.
 * upstreamSemantics != 0: upstream OpenFOAM's fvMatrix::setValuesFromList instead (the reference deviates from it: it uses the
 * TRANSPOSED coefficient in both source sums and clears only the set row's triangle, leaving the other rows coupled to the set
 * cell although their source was corrected -- SURVEY appendix B style quirk): source[nei] -= lower*value for a set owner,
 * source[own] -= upper*value for a set neighbour (face order per row as above), and BOTH triangles cleared at every face that
 * touches a set cell.                                                                                                          */
void orc_set_values(label n, label nf, const label *lo, const label *up, label nSet, const label *cellLabels, const scalar *values,
                    int upstreamSemantics, scalar *psi, const scalar *diag, scalar *source, const scalar *upperIn, const scalar *lowerIn,
                    scalar *upperOut, scalar *lowerOut, int nPatches, const label *patchSizes, const label *const *faceCells,
                    scalar *const *iCoeffs, scalar *const *bCoeffs)
{
    unsigned char *mask = (unsigned char *)calloc((size_t)(n ? n : 1), 1);
    scalar *val = (scalar *)calloc((size_t)(n ? n : 1), sizeof(scalar));
    const scalar *L = lowerIn ? lowerIn : upperIn;
    for (label i = 0; i < nSet; i++) {
        const label c = cellLabels[i];
        psi[c] = values[i];
        source[c] = values[i] * diag[c];
        mask[c] = 1; val[c] = values[i];
    }
    tables t = mk(n, nf, lo, up);
    for (label c = 0; c < n; c++) {
        if (mask[c]) continue;
        scalar out = source[c];
        for (label j = t.os[c]; j < t.os[c + 1]; j++)
            if (mask[up[j]]) out = fma(-(upstreamSemantics ? upperIn[j] : L[j]), val[up[j]], out);
        for (label j = t.ls[c]; j < t.ls[c + 1]; j++) {
            const label g = t.losort[j];
            if (mask[lo[g]]) out = fma(-(upstreamSemantics ? L[g] : upperIn[g]), val[lo[g]], out);
        }
        source[c] = out;
    }
    for (label f = 0; f < nf; f++) {
        const int oSet = mask[lo[f]], nSetF = mask[up[f]];
        upperOut[f] = (upstreamSemantics ? (oSet || nSetF) : oSet) ? 0.0 : upperIn[f];
        lowerOut[f] = (upstreamSemantics ? (oSet || nSetF) : nSetF) ? 0.0 : L[f];
    }
    for (int p = 0; p < nPatches; p++)
        for (label i = 0; i < patchSizes[p]; i++)
            if (mask[faceCells[p][i]]) { iCoeffs[p][i] = 0.0; bCoeffs[p][i] = 0.0; }
    rel(t); free(mask); free(val);
}
