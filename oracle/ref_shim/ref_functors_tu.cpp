// Translation unit that compiles the REFERENCE's device row functors as host code, from where they lie:
//   smoothers/Jacobi/JacobiSmootherF.H (JacobiSmootherFunctor), preconditioners/AINVPreconditioner/AINVPreconditionerF.H
//   (AINVPreconditionerFunctor).  `textures<T>` (a read-only cached view in the reference) becomes a plain pointer view and
//   __device__ expands to nothing.  What this pins is the STRUCTURE of the row arithmetic (which coefficient multiplies which
//   neighbour value, signs, the rD*(r - sum) and (1-w)psi + w rD (b - sum) forms); which mul/add pairs a compiler fuses is
//   its own choice (gcc here, nvcc there), so tests compare within a few ulp of the row magnitude, not bitwise.
#include <cstdint>
typedef int32_t label_t;
namespace Foam
{
typedef int32_t label;
typedef double scalar;
template <class T> struct textures { const T* p; textures(const T* q) : p(q) {} T operator[](label i) const { return p[i]; } };
}
#define __device__
#define __host__
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
#include REF_FILE(smoothers/Jacobi/JacobiSmootherF.H)
#include REF_FILE(preconditioners/AINVPreconditioner/AINVPreconditionerF.H)

extern "C" void ref_jacobi_rows(int n, double omega, const double* psi, const double* diag, const double* b, const double* lower, const double* upper,
                                const int32_t* own, const int32_t* nei, const int32_t* ownStart, const int32_t* losortStart, const int32_t* losort, double* out)
{
    Foam::JacobiSmootherFunctor<false, 3> f(omega, Foam::textures<double>(psi), diag, b, lower, upper, own, nei, ownStart, losortStart, losort);
    for (int c = 0; c < n; c++) out[c] = f(c);
}
extern "C" void ref_ainv_rows(int n, const double* r, const double* rD, const double* lower, const double* upper, const int32_t* own, const int32_t* nei,
                              const int32_t* ownStart, const int32_t* losortStart, const int32_t* losort, double* out)
{
    Foam::AINVPreconditionerFunctor<false, 3> f(Foam::textures<double>(r), Foam::textures<double>(rD), lower, upper, own, nei, ownStart, losortStart, losort);
    for (int c = 0; c < n; c++) out[c] = f(c);
}
