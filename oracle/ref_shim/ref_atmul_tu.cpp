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
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <functional>
#include <tuple>
#include <utility>
#define __device__
#define __host__
#define __HOST____DEVICE__
namespace thrust
{
using std::tuple; using std::get; using std::make_tuple; using std::unary_function; using std::binary_function;
struct counting_iterator
{
    int v;
    int operator*() const { return v; }
    counting_iterator& operator++() { ++v; return *this; }
    counting_iterator operator+(int n) const { return counting_iterator{v + n}; }
    bool operator!=(const counting_iterator& o) const { return v != o.v; }
};
inline counting_iterator make_counting_iterator(int v) { return counting_iterator{v}; }
template <class... P> struct zip_iterator
{
    std::tuple<P...> p;
    template <std::size_t... I> auto deref(std::index_sequence<I...>) const { return std::make_tuple(*std::get<I>(p)...); }
    template <std::size_t... I> void inc(std::index_sequence<I...>) { int d[] = {(++std::get<I>(p), 0)...}; (void)d; }
    auto operator*() const { return deref(std::index_sequence_for<P...>()); }
    zip_iterator& operator++() { inc(std::index_sequence_for<P...>()); return *this; }
};
template <class... P> zip_iterator<P...> make_zip_iterator(const std::tuple<P...>& t) { return zip_iterator<P...>{t}; }
template <class It, class F> struct transform_iterator
{
    It it; mutable F f;
    auto operator*() const { return f(*it); }
    transform_iterator& operator++() { ++it; return *this; }
};
template <class It, class F> transform_iterator<It, F> make_transform_iterator(It it, F f) { return transform_iterator<It, F>{it, f}; }
template <class B, class I> struct permutation_iterator
{
    B base; I idx;
    auto& operator*() const { return base[*idx]; }
    permutation_iterator& operator++() { ++idx; return *this; }
};
template <class B, class I> permutation_iterator<B, I> make_permutation_iterator(B b, I i) { return permutation_iterator<B, I>{b, i}; }
template <class In, class Out, class F> void transform(In first, In last, Out out, F f) { for (; first != last; ++first, ++out) *out = f(*first); }
template <class In, class In2, class Out, class F> void transform(In first, In last, In2 in2, Out out, F f)
{
    for (; first != last; ++first, ++in2, ++out) *out = f(*first, *in2);
}
}
namespace Foam
{
typedef int32_t label; typedef double scalar; typedef unsigned char direction;
#define forAll(list, i) for (Foam::label i = 0; i < (list).size(); i++)
template <class T> struct textures { const T* p; textures(const T* q) : p(q) {} T operator[](const int& i) const { return p[i]; } };
template <class T> class gpuList
{
    T* p_; label n_; bool own_;
public:
    gpuList() : p_(0), n_(0), own_(false) {}
    gpuList(T* p, label n) : p_(p), n_(n), own_(false) {}
    explicit gpuList(label n) : p_((T*)std::calloc(n ? n : 1, sizeof(T))), n_(n), own_(true) {}
    gpuList(label n, const T& t) : p_((T*)std::malloc((n ? n : 1) * sizeof(T))), n_(n), own_(true) { *this = t; }
    gpuList(const gpuList&) = delete;
    ~gpuList() { if (own_) std::free(p_); }
    void setSize(label n) { if (own_) std::free(p_); p_ = (T*)std::calloc(n ? n : 1, sizeof(T)); n_ = n; own_ = true; }
    label size() const { return n_; }
    T* begin() { return p_; } const T* begin() const { return p_; }
    T* end() { return p_ + n_; } const T* end() const { return p_ + n_; }
    T* data() { return p_; } const T* data() const { return p_; }
    void operator=(const T& t) { for (label i = 0; i < n_; i++) p_[i] = t; }
};
template <class T> using gpuField = gpuList<T>;
typedef gpuList<scalar> scalargpuField; typedef gpuList<label> labelgpuList;
template <class T> struct textureBind { const T* d; textureBind(const gpuList<T>& l) : d(l.data()) {} textures<T> operator()() const { return textures<T>(d); } };
template <class T> class tmp
{
    mutable T* p_; bool own_;
public:
    tmp(T* p) : p_(p), own_(true) {}
    tmp(T& r) : p_(&r), own_(false) {}
    tmp(const tmp& t) : p_(t.p_), own_(t.own_) { t.p_ = 0; }
    ~tmp() { if (own_) delete p_; }
    T& operator()() { return *p_; } const T& operator()() const { return *p_; }
    T* release() { T* p = p_; p_ = 0; return p; }
    void clear() const {}
};
inline tmp<scalargpuField> operator-(const scalargpuField& f) { scalargpuField* r = new scalargpuField(f.size()); for (label i = 0; i < f.size(); i++) r->data()[i] = -f.data()[i]; return tmp<scalargpuField>(r); }
template <template <class> class F, class T> struct FieldField
{
    label n; FieldField() : n(0) {} explicit FieldField(label m) : n(m) {}
    label size() const { return n; }
    const F<T>& operator[](label) const { static F<T> e; return e; }
    void set(label, const tmp<F<T> >&) {}
};
struct lduInterfaceFieldPtrsList { label size() const { return 0; } bool set(label) const { return false; } };
class lduAddressing
{
public:
    label n; labelgpuList lower, upper, ownerSort, ownerStart, losortStart, losort, none;
    label size() const { return n; }
    const labelgpuList& lowerAddr() const { return lower; }
    const labelgpuList& upperAddr() const { return upper; }
    const labelgpuList& ownerSortAddr() const { return ownerSort; }
    const labelgpuList& ownerStartAddr() const { return ownerStart; }
    const labelgpuList& losortStartAddr() const { return losortStart; }
    const labelgpuList& losortAddr() const { return losort; }
    const labelgpuList& patchSortCells(label) const { return none; }
    const labelgpuList& patchSortAddr(label) const { return none; }
    const labelgpuList& patchSortStartAddr(label) const { return none; }
};
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
};
// GENERATE_UNARY_OPERATOR_FUNCTORS(-,negate) of fields/Fields/gpuField/gpuFieldM.H:116-124 (that header needs all of gpuField)
template <class Type, class RType> struct negateUnaryOperatorFunctor { RType operator()(const Type& t) { return -t; } };
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
namespace Foam { label lduMatrixSolutionCache::favourSpeed = 0; scalargpuField lduMatrixSolutionCache::first_; scalargpuField lduMatrixSolutionCache::second_; }

// which: 0 Amul, 1 Tmul, 2 residual (in = psi, in2 = source), 3 sumA, 4 H1.  lowerSort[j] = lower[losort[j]] etc. is the caller's
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
    else m.H1(o);
}
