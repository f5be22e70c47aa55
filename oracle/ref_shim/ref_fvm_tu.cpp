// Translation unit that compiles the REFERENCE's fvMatrix-assembly functors as host code and RUNS them (test infrastructure;
// built only where /root/reference exists, oracle/Makefile target _ref/libref_fvm.so).  What comes from the reference, untouched:
//   * its primitives, included IN PLACE: primitives/Vector/vector/vector.H (-> Vector.H, VectorI.H, VectorSpace.H, VectorSpaceI.H,
//     scalar.H, doubleScalar.H, Scalar.H: sign / pos / mag / max / min / component / cmptMax / cmptMin / cmptMag / cmptMultiply,
//     label.H, ops.H, products.H, pTraits.H) and vector.C (vector::zero);
//   * limitedSchemes/LimitedScheme/NVDTVD.H (r) and limitedSchemes/limitedLinear/limitedLinear.H (limiter), included in place;
//   * the functor definitions that live inside otherwise un-compilable .C files, cut out by line range at build time into a scratch
//     directory outside the repository (ref_shim/fvm/extract.sh) and #included from there:
//       fvMatrix.C:36-76       fvMatrixPatchAddFunctor            (addToInternalField / subtractFromInternalField: addBoundaryDiag, addBoundarySource)
//       fvMatrix.C:245-286     fvMatrixAddBoundarySourceFunctor   (coupled part of addBoundarySource)
//       fvMatrix.C:352-452     fvMatrixSetValuesSourceFunctor, fvMatrixSetValuesClearFacesFunctor   (setValues)
//       fvMatrix.C:983-1084    fvMatrixRelaxDiagonalDominanceFunctor, fvMatrixRelaxAddToDiagonalFunctor + its five component functors (relax)
//       fvcSurfaceIntegrate.C:40-132   surfaceIntegrateFunctor<Type, integrate>, surfaceIntegratePatchFunctor
//       gaussGrad.C:31-131     gaussGradFunctor, gaussGradPatchFunctor
//       surfaceInterpolationScheme.C:273-279        surfaceInterpolationSchemeInterpolateFunctor
//       limitedSurfaceInterpolationScheme.C:155-161 limitedSurfaceInterpolationSchemeWeightsFunctor
//       LimitedScheme.C:32-57  LimitedSchemeCalcLimiterFunctor
//       lduMatrixTemplates.C:35-50  lduMatrixfaceHFunctor
// What is scaffolding here (no arithmetic): thrust::tuple / get as std::tuple / get, `word`, the error stream, an Istream that
// holds one scalar (limitedLinearLimiter's constructor reads k from it), and the loops that apply a functor to every row / face /
// patch cell in ascending order -- what thrust::transform does with it.
// Compiled -ffp-contract=fast -march=x86-64-v3: a*b + c inside one functor fuses to one fma, as it does in device code.
#define __device__
#define __host__
#define __HOST____DEVICE__
#define __constant__ static
#include <cstdint>
#include <functional>
#include <tuple>
namespace thrust { using std::tuple; using std::get; }
#include "vector.H"
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_SRC/rel)
#include REF_FILE(OpenFOAM/primitives/Vector/vector/vector.C)
namespace Foam
{
class Istream { public: scalar v; };
scalar readScalar(Istream& is) { return is.v; }
const scalar pTraits<scalar>::zero = 0.0;      // Scalar.C (needs the IOstreams): the one constant the functors use
}
#include "fvMatrix_patchAdd.inc"
#include "fvMatrix_boundarySource.inc"
#include "fvMatrix_setValues.inc"
#include "fvMatrix_relax.inc"
namespace Foam { namespace fvc {
#include "fvcSurfaceIntegrate.inc"
} }
#include "gaussGrad.inc"
namespace Foam {
#include "interpolate.inc"
#include "limitedWeights.inc"
}
#include "calcLimiter.inc"
#include "faceH.inc"
#include "NVDTVD.H"
#include "limitedLinear.H"

using namespace Foam;
typedef int32_t i32;

// ---- per unique patch cell u (cells[u]; its patch faces sort[start[u] .. start[u+1]) ascending): field[cells[u]] = F(field[cells[u]], u)
// kind 0 fvMatrixPatchAddFunctor<scalar,true>, 1 <scalar,false>, 2 fvMatrixAddBoundarySourceFunctor (pf = pbc, q = pnf),
//      3..7 fvMatrixRelaxAddToDiagonalFunctor with componetZero / magComponetZero / maxComponentMagComponet / negativeComponetZero /
//      negativeComponetMin, 8 fvc::surfaceIntegratePatchFunctor
extern "C" void ref_fvm_patch_rows(int kind, int nU, const i32* cells, const i32* start, const i32* sort, const double* pf, const double* q, double* field)
{
    for (int u = 0; u < nU; u++)
    {
        double& d = field[cells[u]];
        switch (kind)
        {
        case 0: d = fvMatrixPatchAddFunctor<scalar, true>(pf, start, sort)(d, u); break;
        case 1: d = fvMatrixPatchAddFunctor<scalar, false>(pf, start, sort)(d, u); break;
        case 2: d = fvMatrixAddBoundarySourceFunctor<scalar>(pf, q, start, sort)(d, u); break;
        case 3: d = fvMatrixRelaxAddToDiagonalFunctor<scalar, componetZeroFunctor<scalar> >(componetZeroFunctor<scalar>(), pf, start, sort)(u, d); break;
        case 4: d = fvMatrixRelaxAddToDiagonalFunctor<scalar, magComponetZeroFunctor<scalar> >(magComponetZeroFunctor<scalar>(), pf, start, sort)(u, d); break;
        case 5: d = fvMatrixRelaxAddToDiagonalFunctor<scalar, maxComponentMagComponetFunctor<scalar> >(maxComponentMagComponetFunctor<scalar>(), pf, start, sort)(u, d); break;
        case 6: d = fvMatrixRelaxAddToDiagonalFunctor<scalar, negativeComponetZeroFunctor<scalar> >(negativeComponetZeroFunctor<scalar>(), pf, start, sort)(u, d); break;
        case 7: d = fvMatrixRelaxAddToDiagonalFunctor<scalar, negativeComponetMinFunctor<scalar> >(negativeComponetMinFunctor<scalar>(), pf, start, sort)(u, d); break;
        case 8: d = fvc::surfaceIntegratePatchFunctor<scalar>(pf, start, sort)(u, d); break;
        }
    }
}
extern "C" void ref_fvm_relax_dominance(int n, const double* sumOff, double* D)
{
    fvMatrixRelaxDiagonalDominanceFunctor<scalar> f;
    for (int c = 0; c < n; c++) D[c] = f(D[c], sumOff[c]);
}
// fvc::surfaceIntegrate's row functor (integrate = 1) / surfaceSum's (integrate = 0)
extern "C" void ref_surface_integrate_rows(int integrate, int n, const double* issf, const i32* ownStart, const i32* neiStart, const i32* own, const i32* nei,
                                           const i32* losort, double* out)
{
    if (integrate) { fvc::surfaceIntegrateFunctor<scalar, true> f(issf, ownStart, neiStart, own, nei, losort); for (int c = 0; c < n; c++) out[c] = f(c); }
    else { fvc::surfaceIntegrateFunctor<scalar, false> f(issf, ownStart, neiStart, own, nei, losort); for (int c = 0; c < n; c++) out[c] = f(c); }
}
extern "C" void ref_face_interpolate(int nFaces, const i32* P, const i32* N, const double* lambda, const double* vf, double* sf)
{
    surfaceInterpolationSchemeInterpolateFunctor<scalar> f;
    for (int i = 0; i < nFaces; i++) sf[i] = f(thrust::tuple<scalar, scalar, scalar>(lambda[i], vf[P[i]], vf[N[i]]));
}
// the same functor on a vector field (interpolate(U), interpolate(grad p)): AoS in, AoS out
extern "C" void ref_face_interpolate_vector(int nFaces, const i32* P, const i32* N, const double* lambda, const double* vf3, double* sf3)
{
    surfaceInterpolationSchemeInterpolateFunctor<vector> f;
    const vector* v = reinterpret_cast<const vector*>(vf3);
    vector* o = reinterpret_cast<vector*>(sf3);
    for (int i = 0; i < nFaces; i++) o[i] = f(thrust::tuple<scalar, vector, vector>(lambda[i], v[P[i]], v[N[i]]));
}
// Sf & vf (the dot product of VectorI.H:129-132) per face -- phi = Sf & interpolate(U)
extern "C" void ref_face_dot(int nFaces, const double* a3, const double* b3, double* out)
{
    const vector* a = reinterpret_cast<const vector*>(a3); const vector* b = reinterpret_cast<const vector*>(b3);
    for (int i = 0; i < nFaces; i++) out[i] = a[i] & b[i];
}
// gaussGrad<scalar>::gradf: internal faces (rows), then per patch (patch rows); Sf AoS, out AoS
extern "C" void ref_gauss_grad_rows(int n, const double* Sf3, const double* issf, const i32* ownStart, const i32* neiStart, const i32* own, const i32* nei,
                                    const i32* losort, double* out3)
{
    gaussGradFunctor<scalar, vector> f(vector::zero, reinterpret_cast<const vector*>(Sf3), issf, ownStart, neiStart, own, nei, losort);
    vector* o = reinterpret_cast<vector*>(out3);
    for (int c = 0; c < n; c++) o[c] = f(c);
}
extern "C" void ref_gauss_grad_patch_rows(int nU, const i32* cells, const i32* start, const i32* sort, const double* pSf3, const double* pssf, double* grad3)
{
    gaussGradPatchFunctor<scalar, vector> f(reinterpret_cast<const vector*>(pSf3), pssf, start, sort);
    vector* g = reinterpret_cast<vector*>(grad3);
    for (int u = 0; u < nU; u++) g[cells[u]] = f(u, g[cells[u]]);
}
extern "C" void ref_faceH(int nFaces, const i32* l, const i32* u, const double* Lower, const double* Upper, const double* psi, double* out)
{
    lduMatrixfaceHFunctor<scalar> f;
    for (int i = 0; i < nFaces; i++) out[i] = f(thrust::tuple<scalar, scalar, scalar, scalar>(Upper[i], psi[u[i]], Lower[i], psi[l[i]]));
}
// limitedLinear(k): LimitedScheme::calcLimiter's functor (limiter = limitedLinearLimiter<NVDTVD>) per internal face, then
// limitedSurfaceInterpolationScheme::weights' functor.  gradc, C: AoS.
extern "C" void ref_limited_linear(int nFaces, const i32* owner, const i32* neighbour, double k, const double* CDweights, const double* faceFlux,
                                   const double* lPhi, const double* gradc3, const double* C3, double* limiterOut, double* weightsOut)
{
    Istream is; is.v = k;
    limitedLinearLimiter<NVDTVD> lim(is);
    LimitedSchemeCalcLimiterFunctor<limitedLinearLimiter<NVDTVD>, scalar, vector> f(lim);
    limitedSurfaceInterpolationSchemeWeightsFunctor w;
    const vector* g = reinterpret_cast<const vector*>(gradc3); const vector* C = reinterpret_cast<const vector*>(C3);
    for (int i = 0; i < nFaces; i++)
    {
        const i32 P = owner[i], N = neighbour[i];
        limiterOut[i] = f(CDweights[i], thrust::tuple<scalar, scalar, scalar, vector, vector, vector, vector>(faceFlux[i], lPhi[P], lPhi[N], g[P], g[N], C[N], C[P]));
        weightsOut[i] = w(thrust::tuple<scalar, scalar, scalar>(limiterOut[i], CDweights[i], faceFlux[i]));
    }
}
// fvMatrix::setValuesFromList's source functor over every row, and the clear-faces functor over upper / lower
// (masks as fvMatrix.C:497-551 builds them: ownMask[f] = cellMask[own[f]], neiMask[f] = cellMask[nei[f]])
extern "C" void ref_set_values_source(int n, int nFaces, const i32* own, const i32* nei, const i32* ownStart, const i32* neiStart, const i32* losort,
                                      const unsigned char* cellMask, const double* cellValues, const double* Upper, const double* Lower, double* source,
                                      double* upperOut, double* lowerOutOrNull)
{
    bool* ownMask = new bool[nFaces ? nFaces : 1]; bool* neiMask = new bool[nFaces ? nFaces : 1];
    for (int i = 0; i < nFaces; i++) { ownMask[i] = cellMask[own[i]] != 0; neiMask[i] = cellMask[nei[i]] != 0; }
    fvMatrixSetValuesSourceFunctor<scalar> f(ownMask, neiMask, cellValues, Upper, Lower, ownStart, neiStart, own, nei, losort);
    for (int c = 0; c < n; c++) source[c] = f(source[c], thrust::tuple<label, bool>(c, cellMask[c] != 0));
    fvMatrixSetValuesClearFacesFunctor<scalar> z(0.0);
    for (int i = 0; i < nFaces; i++) upperOut[i] = z(Upper[i], ownMask[i]);
    if (lowerOutOrNull) for (int i = 0; i < nFaces; i++) lowerOutOrNull[i] = z(Lower[i], neiMask[i]);
    delete[] ownMask; delete[] neiMask;
}
