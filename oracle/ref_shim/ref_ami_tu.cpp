// Translation unit that compiles the REFERENCE's AMI interpolation functors as host code, from where they lie:
//   src/meshTools/AMIInterpolation/AMIInterpolation/AMIInterpolationF.H   (AMIInterpolationInterpolateFunctor / ...NoCorrectionFunctor)
//   src/OpenFOAM/primitives/ops/ops.H                                    (plusEqOp, multiplyWeightedOp: cop_(x, weight*y))
// driven as AMIInterpolation.C:1694-1806 (interpolateToSource / interpolateToTarget on gpuLists) does: one functor call per
// face over [addressStart[face], addressStart[face+1]).  Pins the oracle's cyclicAMI neighbour values (ldu_oracle.c
// update_interfaces): address order, weight-times-value pairing, the low-weight default, the contraction of out += w*f.
#include <cmath>
#include <cstdint>
#include <cstdlib>
namespace Foam
{
typedef int32_t label;
typedef double scalar;
template <class T> struct pTraits;
template <> struct pTraits<double> { static constexpr double zero = 0.0; };
}
#define __HOST____DEVICE__
#define __host__
#define __device__
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_SRC/rel)
#include REF_FILE(OpenFOAM/primitives/ops/ops.H)
#include REF_FILE(meshTools/AMIInterpolation/AMIInterpolation/AMIInterpolationF.H)

extern "C" void ref_ami_interpolate(int n, const int32_t* start, const int32_t* address, const double* weights, const double* fld,
                                    double lowWeightCorrection, const double* weightsSum, const double* defaultValues, double* out)
{
    using namespace Foam;
    typedef multiplyWeightedOp<scalar, plusEqOp<scalar> > Cop;
    plusEqOp<scalar> peq;
    Cop cop(peq);
    if (lowWeightCorrection > 0) {
        AMIInterpolationInterpolateFunctor<scalar, Cop> f(lowWeightCorrection, cop, defaultValues, fld, address, start, weights, weightsSum);
        for (label i = 0; i < n; ++i) out[i] = f(i);
    } else {
        AMIInterpolationInterpolateNoCorrectionFunctor<scalar, Cop> f(cop, fld, address, start, weights);
        for (label i = 0; i < n; ++i) out[i] = f(i);
    }
}
