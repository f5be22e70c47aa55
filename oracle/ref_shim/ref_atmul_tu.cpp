// Translation unit that compiles the REFERENCE's lduMatrix/lduMatrixATmul.C where it lies and RUNS it on the host:
//   lduMatrix::Amul / Tmul (callMultiply + matrixMultiplyFunctor<fast,3>), sumA, residual, H1,
// together with lduAddressing/lduAddressingFunctors.H (matrixOperation, matrixFastOperation and their row functors),
// lduMatrix/lduMatrixFunctors.H and primitives/ops/ops.H, all included from /root/reference.  What this file supplies is the
// scaffolding those sources expect: a minimal host `thrust` (counting / zip / transform / permutation iterators over raw
// pointers and a sequential transform), gpuList as a pointer view, tmp<>, and an lduMatrix / lduAddressing pair that only
// holds the caller's arrays.  Interfaces are empty (the coupled update is pinned through the solver sources).  As in
// ref_functors_tu.cpp, `textures<T>` is a pointer view and the mul/add contraction is the host compiler's choice, so tests
// compare within a few ulp of the row magnitude; the face order, the coefficient/neighbour pairing, the signs and the
// fast path's handling of rows with more than three neighbour-side faces are the reference's own.
#include "foam_host_shim.H"
namespace Foam
{
class lduMatrix
{
public:
    lduAddressing addr; scalargpuField Diag, Lower, Upper, LowerSort, UpperSort; label level_; bool coarsest_;
    const scalargpuField* lowerPtr_; const scalargpuField* upperPtr_;
    const lduAddressing& lduAddr() const { return addr; }
    const scalargpuField& diag() const { return Diag; }
    const scalargpuField& lower() const { return Lower; }
    const scalargpuField& upper() const { return Upper; }
    const scalargpuField& lowerSort() const { return LowerSort; }
    const scalargpuField& upperSort() const { return UpperSort; }
    label level() const { return level_; }
    bool coarsestLevel() const { return coarsest_; }
    void initMatrixInterfaces(const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&, const scalargpuField&, scalargpuField&, const direction) const {}
    void updateMatrixInterfaces(const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&, const scalargpuField&, scalargpuField&, const direction) const {}
    void Amul(scalargpuField&, const tmp<scalargpuField>&, const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&, const direction) const;
    void Tmul(scalargpuField&, const tmp<scalargpuField>&, const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&, const direction) const;
    void sumA(scalargpuField&, const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&) const;
    void residual(scalargpuField&, const scalargpuField&, const scalargpuField&, const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&, const direction) const;
    tmp<scalargpuField> residual(const scalargpuField&, const scalargpuField&, const FieldField<gpuField, scalar>&, const lduInterfaceFieldPtrsList&, const direction) const;
    void H1(scalargpuField&) const;
    tmp<scalargpuField> H1() const;
    // what preconditioners/AINVPreconditioner/AINVPreconditioner.{H,C} need of lduMatrix.H:99-269,438-520
    class solver { const lduMatrix& m_; public: solver(const lduMatrix& m) : m_(m) {} const lduMatrix& matrix() const { return m_; } };
    class preconditioner
    {
    protected:
        const solver& solver_;
    public:
        preconditioner(const solver& s) : solver_(s) {}
        virtual ~preconditioner() {}
        virtual void precondition(scalargpuField&, const scalargpuField&, const direction = 0) const = 0;
        virtual void preconditionT(scalargpuField&, const scalargpuField&, const direction = 0) const = 0;
        template <class T> struct addsymMatrixConstructorToTable {};
        template <class T> struct addasymMatrixConstructorToTable {};
    };
};
class dictionary {};
#define TypeName(name) static const char* typeName_() { return name; }
#define defineTypeNameAndDebug(T, d)
// GENERATE_OPERATOR_FUNCTORS(/,divide) of fields/Fields/gpuField/gpuFieldM.H:151-159 (scalar op field)
template <class T1, class T2, class R> struct divideOperatorSFFunctor { const T1 t1; divideOperatorSFFunctor(T1 t) : t1(t) {} R operator()(const T2& t2) { return t1 / t2; } };
}
#define lduMatrix_H
#define lduAddressing_H
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
#include REF_FILE(../../primitives/ops/ops.H)
#include REF_FILE(lduAddressing/lduAddressingFunctors.H)
#include REF_FILE(lduMatrix/lduMatrixFunctors.H)
#include REF_FILE(lduMatrix/lduMatrixATmul.C)
#include REF_FILE(preconditioners/AINVPreconditioner/AINVPreconditioner.C)
namespace Foam { label lduMatrixSolutionCache::favourSpeed = 0; scalargpuField lduMatrixSolutionCache::first_; scalargpuField lduMatrixSolutionCache::second_; }

// which: 0 Amul, 1 Tmul, 2 residual (in = psi, in2 = source), 3 sumA, 4 H1, 5 / 6 AINVPreconditioner::precondition / preconditionT (in = r).  lowerSort[j] = lower[losort[j]] etc. is the caller's
// (lduMatrix::lowerSort(), lduMatrix.C); favourSpeed / level / coarsest select the reference's fast paths.
extern "C" void ref_atmul(int which, int favourSpeed, int level, int coarsest, int n, int nFaces, const int32_t* lower, const int32_t* upper, const int32_t* ownerSort,
                          const int32_t* ownerStart, const int32_t* losortStart, const int32_t* losort, const double* diag, const double* lowerC, const double* upperC,
                          const double* lowerSortC, const double* upperSortC, const double* in, const double* in2, double* out)
{
    using namespace Foam;
    typedef gpuList<label> L; typedef gpuList<scalar> S;
    lduMatrix m;
    m.addr.n = n;
    new (&m.addr.lower) L((label*)lower, nFaces); new (&m.addr.upper) L((label*)upper, nFaces); new (&m.addr.ownerSort) L((label*)ownerSort, nFaces);
    new (&m.addr.ownerStart) L((label*)ownerStart, n + 1); new (&m.addr.losortStart) L((label*)losortStart, n + 1); new (&m.addr.losort) L((label*)losort, nFaces);
    new (&m.Diag) S((scalar*)diag, n); new (&m.Lower) S((scalar*)lowerC, nFaces); new (&m.Upper) S((scalar*)upperC, nFaces);
    new (&m.LowerSort) S((scalar*)lowerSortC, nFaces); new (&m.UpperSort) S((scalar*)upperSortC, nFaces);
    m.level_ = level; m.coarsest_ = coarsest != 0; m.lowerPtr_ = &m.Lower; m.upperPtr_ = &m.Upper;
    lduMatrixSolutionCache::favourSpeed = favourSpeed;
    FieldField<gpuField, scalar> noCoeffs; lduInterfaceFieldPtrsList noInterfaces;
    S o(out, n), x((scalar*)in, n), b((scalar*)in2, n);
    if (which == 0) m.Amul(o, tmp<S>(x), noCoeffs, noInterfaces, 0);
    else if (which == 1) m.Tmul(o, tmp<S>(x), noCoeffs, noInterfaces, 0);
    else if (which == 2) m.residual(o, x, b, noCoeffs, noInterfaces, 0);
    else if (which == 3) m.sumA(o, noCoeffs, noInterfaces);
    else if (which == 4) m.H1(o);
    else
    {
        lduMatrix::solver sol(m); dictionary dict;
        AINVPreconditioner P(sol, dict);            // rD = 1/diag into the solution cache (AINVPreconditioner.C:18-41)
        if (which == 5) P.precondition(o, x); else P.preconditionT(o, x);
    }
}
