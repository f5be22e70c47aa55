// Translation unit that compiles the REFERENCE's lduMatrix/lduMatrixOperations.C where it lies and RUNS it on the host:
//   sumDiag, negSumDiag, sumMagOffDiag, H (both paths), operator*=(field) -- with lduAddressingFunctors.H and ops.H from
//   /root/reference as in ref_atmul_tu.cpp.  The lduMatrix below only stores coefficients the way the reference's does
//   (lowerPtr_ / diagPtr_ / upperPtr_, lower() of a symmetric matrix is its upper()); operator= / negate / += / -= of that
//   file are compiled too and run on the same storage.
#include "foam_host_shim.H"
#include <iostream>
namespace Foam
{
struct errorT { template <class T> errorT& operator<<(const T&) { return *this; } };
static errorT FatalError;
inline const char* abort(errorT&) { std::abort(); return ""; }
#define WarningIn(x) FatalError
static const char nl = '\n';
inline errorT& endl(errorT& e) { return e; }
inline errorT& operator<<(errorT& e, errorT& (*)(errorT&)) { return e; }
inline scalar mag(const scalar x) { return std::fabs(x); }
// GENERATE_UNARY_FUNCTION_FUNCTORS(mag) and GENERATE_OPERATOR_FUNCTORS(*,multiply) of fields/Fields/gpuField/gpuFieldM.H:127-175
template <class Type, class RType> struct magUnaryFunctionFunctor { RType operator()(const Type& t) { return mag(t); } };
template <class T1, class T2, class R> struct multiplyOperatorFunctor { R operator()(const T1& a, const T2& b) { return a * b; } };
class lduMatrix
{
public:
    static int debug;
    lduAddressing addr;
    scalargpuField *lowerPtr_, *diagPtr_, *upperPtr_;
    mutable scalargpuField *lowerSortPtr_, *upperSortPtr_;
    lduMatrix() : lowerPtr_(0), diagPtr_(0), upperPtr_(0), lowerSortPtr_(0), upperSortPtr_(0) {}
    const lduAddressing& lduAddr() const { return addr; }
    // lduMatrix.C:221-345: lower() of a symmetric matrix is upper() (and the other way round); the non-const forms create the
    // missing triangle as a copy
    scalargpuField& diag() { return *diagPtr_; }
    const scalargpuField& diag() const { return *diagPtr_; }
    scalargpuField& upper() { if (!upperPtr_) { upperPtr_ = new scalargpuField(lowerPtr_->size()); *upperPtr_ = *lowerPtr_; } return *upperPtr_; }
    scalargpuField& lower() { if (!lowerPtr_) { lowerPtr_ = new scalargpuField(upperPtr_->size()); *lowerPtr_ = *upperPtr_; } return *lowerPtr_; }
    const scalargpuField& upper() const { return upperPtr_ ? *upperPtr_ : *lowerPtr_; }
    const scalargpuField& lower() const { return lowerPtr_ ? *lowerPtr_ : *upperPtr_; }
    const scalargpuField& lowerSort() const     // lduMatrix::calcSortCoeffs: lowerSort[j] = lower[losort[j]]
    {
        if (!lowerSortPtr_) { const scalargpuField& L = lower(); lowerSortPtr_ = new scalargpuField(L.size()); for (label j = 0; j < L.size(); j++) lowerSortPtr_->data()[j] = L.data()[addr.losort.data()[j]]; }
        return *lowerSortPtr_;
    }
    bool diagonal() const { return diagPtr_ && !lowerPtr_ && !upperPtr_; }
    bool symmetric() const { return diagPtr_ && (!lowerPtr_ && upperPtr_); }
    bool asymmetric() const { return diagPtr_ && lowerPtr_ && upperPtr_; }
    void sumDiag(); void negSumDiag(); void sumMagOffDiag(scalargpuField&) const;
    template <class Type> void H(gpuField<Type>&, const gpuField<Type>&) const;
    template <class Type> tmp<gpuField<Type> > H(const gpuField<Type>&) const;
    void operator=(const lduMatrix&); void negate(); void operator+=(const lduMatrix&); void operator-=(const lduMatrix&);
    void operator*=(const scalargpuField&); void operator*=(scalar);
};
int lduMatrix::debug = 0;
}
#define lduMatrix_H
#define lduAddressing_H
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
#include REF_FILE(../../primitives/ops/ops.H)
#include REF_FILE(lduAddressing/lduAddressingFunctors.H)
#include REF_FILE(lduMatrix/lduMatrixOperations.C)
namespace Foam { label lduMatrixSolutionCache::favourSpeed = 0; scalargpuField lduMatrixSolutionCache::first_; scalargpuField lduMatrixSolutionCache::second_; }

// which: 0 sumDiag -> out[n] = diag, 1 negSumDiag -> diag, 2 sumMagOffDiag -> out[n], 3 H(psi) -> out[n], 4 operator*=(sf = in) -> outDiag/outUpper/outLower,
//        5 operator+=(B) then negate(), B = the same matrix scaled by 0.5 via operator= and operator*=(scalar) -> outDiag/outUpper/outLower
extern "C" void ref_ldu_ops(int which, int favourSpeed, int n, int nFaces, const int32_t* lower, const int32_t* upper, const int32_t* ownerSort, const int32_t* ownerStart,
                            const int32_t* losortStart, const int32_t* losort, const double* diag, const double* lowerC, const double* upperC, const double* in,
                            double* out, double* outUpper, double* outLower)
{
    using namespace Foam;
    typedef gpuList<label> L; typedef gpuList<scalar> S;
    lduMatrix m;
    m.addr.n = n;
    new (&m.addr.lower) L((label*)lower, nFaces); new (&m.addr.upper) L((label*)upper, nFaces); new (&m.addr.ownerSort) L((label*)ownerSort, nFaces);
    new (&m.addr.ownerStart) L((label*)ownerStart, n + 1); new (&m.addr.losortStart) L((label*)losortStart, n + 1); new (&m.addr.losort) L((label*)losort, nFaces);
    S d0((scalar*)diag, n), u0((scalar*)upperC, nFaces);
    m.diagPtr_ = new S(n); *m.diagPtr_ = d0;
    m.upperPtr_ = new S(nFaces); *m.upperPtr_ = u0;
    if (lowerC) { S l0((scalar*)lowerC, nFaces); m.lowerPtr_ = new S(nFaces); *m.lowerPtr_ = l0; }
    lduMatrixSolutionCache::favourSpeed = favourSpeed;
    S o(out, n), x((scalar*)in, n);
    if (which == 0) { m.sumDiag(); o = m.diag(); }
    else if (which == 1) { m.negSumDiag(); o = m.diag(); }
    else if (which == 2) { o = 0.0; m.sumMagOffDiag(o); }
    else if (which == 3) m.H(o, x);
    else
    {
        if (which == 4) m *= x;
        else
        {
            lduMatrix B; B.addr.n = n; B.diagPtr_ = new S(n); B.upperPtr_ = new S(nFaces); if (lowerC) B.lowerPtr_ = new S(nFaces);
            B = m; B *= 0.5; m += B; m.negate();
            delete B.diagPtr_; delete B.upperPtr_; delete B.lowerPtr_;
        }
        S ou(outUpper, nFaces), ol(outLower, nFaces);
        o = m.diag(); ou = m.upper(); ol = const_cast<const lduMatrix&>(m).lower();
    }
    delete m.diagPtr_; delete m.upperPtr_; delete m.lowerPtr_; delete m.lowerSortPtr_;
}
