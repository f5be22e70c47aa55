// Translation unit that compiles the REFERENCE's GAMG inter-level functors as host code, from where they lie:
//   solvers/GAMG/GAMGSolverAgglomerateMatrixF.H            (symAgglomerate, diagSymAgglomerate, asymAgglomerate, diagAsymAgglomerate)
//   solvers/GAMG/GAMGAgglomerations/GAMGAgglomeration/GAMGAgglomerationF.H   (restrict, prolong)
// and drives them the way GAMGSolverAgglomerateMatrix.C:218-317 / GAMGAgglomerationTemplates.C:35-170 do: one call per
// unique target of the STABLY sorted restrict addressing (createSort / createTarget, GAMGAgglomerateLduAddressing.C:37-120),
// the functor adding its segment onto the value already there.  thrust::tuple is std::tuple; the atomic variants are
// device-only members that the host never calls.
#include <cstdint>
#include <tuple>
namespace thrust { using std::tuple; using std::get; using std::make_tuple; }
namespace Foam
{
typedef int32_t label;
typedef double scalar;
template <class T> struct pTraits;
template <> struct pTraits<double> { static constexpr double zero = 0.0; };
}
template <class T> inline void atomicAdd(T*, T) {}
#define __host__
#define __device__
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
#include REF_FILE(solvers/GAMG/GAMGSolverAgglomerateMatrixF.H)
#include REF_FILE(solvers/GAMG/GAMGAgglomerations/GAMGAgglomeration/GAMGAgglomerationF.H)

using namespace Foam;
// segments: target[s] (coarse face >= 0, or -(coarse cell)-1), fine faces sort[start[s] .. start[s+1])
extern "C" void ref_gamg_agglomerate_matrix(int asym, int nSeg, const int32_t* target, const int32_t* start, const int32_t* sort,
                                            const double* fineUpper, const double* fineLower, const bool* flip,
                                            double* coarseDiag, double* coarseUpper, double* coarseLower)
{
    for (int s = 0; s < nSeg; ++s) {
        const auto seg = thrust::make_tuple(start[s], start[s + 1]);
        const label t = target[s];
        if (t >= 0) {
            if (asym) {
                GAMG::asymAgglomerate f(fineUpper, fineLower, flip, sort);
                const thrust::tuple<scalar, scalar> r = f(thrust::make_tuple(coarseUpper[t], coarseLower[t]), seg);
                coarseUpper[t] = thrust::get<0>(r); coarseLower[t] = thrust::get<1>(r);
            } else coarseUpper[t] = GAMG::symAgglomerate(fineUpper, sort)(coarseUpper[t], seg);
        } else {
            const label c = -1 - t;
            if (asym) coarseDiag[c] = GAMG::diagAsymAgglomerate(fineUpper, fineLower, sort)(coarseDiag[c], seg);
            else coarseDiag[c] = GAMG::diagSymAgglomerate(fineUpper, sort)(coarseDiag[c], seg);
        }
    }
}
extern "C" void ref_gamg_restrict(int nSeg, const int32_t* target, const int32_t* start, const int32_t* sort, const double* ff, double* cf)
{
    GAMG::restrict<scalar> f(ff, sort);
    for (int s = 0; s < nSeg; ++s) cf[target[s]] = f(start[s], start[s + 1]);
}
extern "C" void ref_gamg_prolong(int nSeg, const int32_t* target, const int32_t* start, const int32_t* sort, const double* cf, double* ff)
{
    GAMG::prolong<scalar> f(ff, cf, sort, target, start);
    for (label s = 0; s < nSeg; ++s) f(s);
}
