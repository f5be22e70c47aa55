// Translation unit that compiles the REFERENCE's solvers/GAMG/GAMGSolverScale.C where it lies, for its two functors:
//   GAMGSolverScaleFunctor   field = sf*field + (source - sf*Acf)/D      (the pointwise update of GAMGSolver::scale)
//   multiplyTupleFunctor     the term of the two scaling sums
// GAMGSolver::scale itself is Thrust orchestration (zip / transform iterators, thrust::reduce with an unspecified summation
// order); here it only has to COMPILE, against do-nothing stand-ins, so that the functors come from the reference's text.
// Compiled with -ffp-contract=fast like the other pins: `sf*field + (...)` and `source - sf*Acf` contract as nvcc contracts them.
#include <cmath>
#include <cstdint>
#include <functional>
#include <iostream>
#include <tuple>
namespace thrust
{
using std::tuple; using std::get; using std::make_tuple;
template <class T> struct plus { T operator()(const T& a, const T& b) const { return a + b; } };
template <class... A> int make_zip_iterator(A&&...) { return 0; }
template <class... A> int make_transform_iterator(A&&...) { return 0; }
template <class... A> double reduce(A&&...) { return 0.0; }
template <class... A> void transform(A&&...) {}
}
#define __HOST____DEVICE__
namespace Foam
{
typedef int32_t label; typedef double scalar; typedef unsigned char direction;
const scalar VSMALL = 1e-300;
inline scalar stabilise(const scalar x, const scalar y) { return x < 0 ? x - y : x + y; }   // Scalar.H:295-305
static std::ostream& Pout = std::cout;
struct vector2D { scalar v[2]; vector2D(scalar a, scalar b) { v[0] = a; v[1] = b; } scalar x() const { return v[0]; } scalar y() const { return v[1]; } };
template <class T> struct sumOp {};
template <class T> class gpuField {};
template <template <class> class F, class T> class FieldField {};
class lduInterfaceFieldPtrsList {};
struct scalargpuField { const scalar* begin() const { return 0; } const scalar* end() const { return 0; } scalar* begin() { return 0; } scalar* end() { return 0; } };
struct lduMesh { template <class T, class Op> void reduce(T&, const Op&) const {} };
struct lduMatrix
{
    lduMesh m_; scalargpuField d_;
    const lduMesh& mesh() const { return m_; }
    const scalargpuField& diag() const { return d_; }
    void Amul(scalargpuField&, const scalargpuField&, const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&, const direction) const {}
};
class GAMGSolver
{
public:
    static int debug;
    void scale(scalargpuField& field, scalargpuField& Acf, const lduMatrix& A, const FieldField<gpuField, scalar>& interfaceLevelBouCoeffs,
               const lduInterfaceFieldPtrsList& interfaceLevel, const scalargpuField& source, const direction cmpt) const;
};
int GAMGSolver::debug = 0;
}
#define GAMGSolver_H
#define vector2D_H
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
#include REF_FILE(solvers/GAMG/GAMGSolverScale.C)

extern "C" void ref_gamg_scale_pointwise(int n, double sf, const double* field, const double* source, const double* Acf, const double* D, double* out)
{
    Foam::GAMGSolverScaleFunctor f(sf);
    for (int i = 0; i < n; ++i) out[i] = f(field[i], std::make_tuple(source[i], Acf[i], D[i]));
}
extern "C" void ref_gamg_scale_terms(int n, const double* a, const double* b, double* out)
{
    Foam::multiplyTupleFunctor f;
    for (int i = 0; i < n; ++i) out[i] = f(std::make_tuple(a[i], b[i]));
}
