// miFoam.C -- see miFoam.H.  Host code only; every number is produced by the HIP engine behind
// the C ABI (include/mi_ldu.h).  Compiled with hipcc only because device memory is managed here.
#include "miFoam.H"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstdio>
#include <ctime>

namespace Foam
{

std::ostream& Info = std::cout;

void FatalErrorIn(const std::string& where, const std::string& msg)
{
    // the reference prints "--> FOAM FATAL ERROR" and aborts (error.C); a library throws instead
    throw error("--> FOAM FATAL ERROR:\n" + msg + "\n\n    From function " + where);
}

void miCheck(int rc, const char* where)
{
    if (rc != MI_OK) FatalErrorIn(where, std::string("MI355X engine error ") + std::to_string(rc) + ": " + mi_last_error());
}

#define FOAM_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) FatalErrorIn(#expr, hipGetErrorString(e_)); } while (0)

// one engine context per process = per MPI rank = per GPU (argList.C:775-811 `-device N`)
struct miEngine
{
    mi_ctx_t ctx;
    miEngine() : ctx(nullptr)
    {
        const char* d = std::getenv("MI_DEVICE");
        miCheck(mi_ctx_create(d ? std::atoi(d) : 0, nullptr, &ctx), "miEngine::miEngine()");
    }
    ~miEngine() { mi_ctx_destroy(ctx); }
    static miEngine& New() { static miEngine e; return e; }
};

void* miDeviceAlloc(std::size_t bytes) { miEngine::New(); void* p = nullptr; FOAM_HIP(hipMalloc(&p, bytes)); return p; }
void miDeviceFree(void* p) { (void)hipFree(p); }
void miCopyH2D(void* d, const void* s, std::size_t n) { FOAM_HIP(hipMemcpy(d, s, n, hipMemcpyHostToDevice)); }
void miCopyD2H(void* d, const void* s, std::size_t n) { mi_ctx_synchronize(miEngine::New().ctx); FOAM_HIP(hipMemcpy(d, s, n, hipMemcpyDeviceToHost)); }
void miCopyD2D(void* d, const void* s, std::size_t n) { mi_ctx_synchronize(miEngine::New().ctx); FOAM_HIP(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice)); }
void miDeviceZero(void* p, std::size_t n) { FOAM_HIP(hipMemset(p, 0, n)); }

// ---- solverPerformance ------------------------------------------------------------------------------
const scalar solverPerformance::great_ = 1e20;
const scalar solverPerformance::small_ = 1e-20;
const scalar solverPerformance::vsmall_ = 1e-300;

bool solverPerformance::checkConvergence(scalar tol, scalar relTol)
{
    converged_ = (finalResidual_ < tol) || (relTol > small_ && finalResidual_ < relTol * initialResidual_);
    return converged_;
}
bool solverPerformance::checkSingularity(scalar residual) { singular_ = residual < vsmall_; return singular_; }
void solverPerformance::print(std::ostream& os) const
{
    os << solverName_ << ":  Solving for " << fieldName_;
    if (singular_) os << ":  solution singularity" << std::endl;
    else os << ", Initial residual = " << initialResidual_ << ", Final residual = " << finalResidual_
            << ", No Iterations " << noIterations_ << std::endl;
}

// ---- lduAddressing --------------------------------------------------------------------------------------
lduAddressing::lduAddressing(label nCells, const labelList& lower, const labelList& upper, const std::vector<labelList>& patchAddr)
: size_(nCells), lower_(lower), upper_(upper), patchAddr_(patchAddr), addr_(nullptr), gamg_(nullptr), gamgCoarsest_(-1)
{
    if (lower_.size() != upper_.size()) FatalErrorIn("lduAddressing::lduAddressing", "lowerAddr and upperAddr differ in size");
    for (const labelList& fc : patchAddr_) { lduInterface i; i.faceCells = fc; interfaces_.push_back(i); }
}
lduAddressing::lduAddressing(label nCells, const labelList& lower, const labelList& upper, const std::vector<lduInterface>& interfaces)
: size_(nCells), lower_(lower), upper_(upper), interfaces_(interfaces), addr_(nullptr), gamg_(nullptr), gamgCoarsest_(-1)
{
    if (lower_.size() != upper_.size()) FatalErrorIn("lduAddressing::lduAddressing", "lowerAddr and upperAddr differ in size");
    for (const lduInterface& i : interfaces_) patchAddr_.push_back(i.faceCells);
    for (std::size_t p = 0; p < interfaces_.size(); ++p) {
        const lduInterface& i = interfaces_[p];
        if (i.type == "cyclic" && (i.neighbPatchID < 0 || i.neighbPatchID >= (label)interfaces_.size() ||
                                    interfaces_[i.neighbPatchID].faceCells.size() != i.faceCells.size()))
            FatalErrorIn("lduAddressing::lduAddressing", "cyclic patch without a matching neighbour patch");
        if (i.type == "cyclicAMI" && (i.neighbPatchID < 0 || i.neighbPatchID >= (label)interfaces_.size() || i.amiStart.size() != i.faceCells.size() + 1))
            FatalErrorIn("lduAddressing::lduAddressing", "cyclicAMI patch without a neighbour patch / AMI addressing");
        if (i.type != "cyclic" && i.type != "processor" && i.type != "coupled" && i.type != "cyclicAMI") FatalErrorIn("lduAddressing::lduAddressing", "Unknown interface type " + i.type);
    }
}
lduAddressing::~lduAddressing() { if (gamg_) mi_gamg_destroy(gamg_); if (addr_) mi_addr_destroy(addr_); }
mi_addr_t lduAddressing::handle() const
{
    if (!addr_) {
        std::vector<label> sizes; std::vector<const label*> ptrs, nbrs;
        for (std::size_t p = 0; p < patchAddr_.size(); ++p) {
            sizes.push_back((label)patchAddr_[p].size()); ptrs.push_back(patchAddr_[p].data());
            const bool plainCyclic = interfaces_[p].type == "cyclic" && !interfaces_[p].transforms;
            nbrs.push_back(plainCyclic ? patchAddr_[(std::size_t)interfaces_[p].neighbPatchID].data() : nullptr);
        }
        miCheck(mi_addr_create_coupled(miEngine::New().ctx, size_, (label)lower_.size(), lower_.data(), upper_.data(), (label)sizes.size(),
                                       sizes.data(), ptrs.data(), nbrs.data(), &addr_), "lduAddressing::handle()");
        for (std::size_t p = 0; p < interfaces_.size(); ++p) {
            const lduInterface& i = interfaces_[p];
            if (i.type == "cyclicAMI") {
                miCheck(mi_addr_set_ami_patch(addr_, (label)p, i.neighbPatchID, i.amiStart.data(), i.amiAddress.data(), i.amiWeights.data(),
                                              i.amiLowWeight.empty() ? nullptr : i.amiLowWeight.data()), "cyclicAMILduInterface");
                if (!i.amiMagSf.empty()) miCheck(mi_addr_set_ami_face_areas(addr_, (label)p, i.amiMagSf.data()), "cyclicAMILduInterface");
            } else if (i.type == "cyclic" && i.transforms)
                miCheck(mi_addr_set_ami_patch(addr_, (label)p, i.neighbPatchID, nullptr, nullptr, nullptr, nullptr), "cyclicLduInterface (transformed)");
        }
    }
    return addr_;
}
mi_gamg_t lduAddressing::dummyAgglomeration(label nLevels) const
{   // agglomerator dummy (dummyAgglomeration.C:45-90): cached like the pair agglomerations, keyed by -nLevels
    if (gamg_ && gamgCoarsest_ != -nLevels) { mi_gamg_destroy(gamg_); gamg_ = nullptr; }
    if (!gamg_) { miCheck(mi_gamg_create_dummy(handle(), nLevels, &gamg_), "dummyAgglomeration::dummyAgglomeration"); gamgCoarsest_ = -nLevels; gamgMerge_ = 1; }
    return gamg_;
}
mi_gamg_t lduAddressing::agglomeration(const scalarField& w, label nCoarsest, label mergeLevels) const
{
    if (gamg_ && (gamgCoarsest_ != nCoarsest || gamgMerge_ != mergeLevels)) { mi_gamg_destroy(gamg_); gamg_ = nullptr; }
    if (!gamg_) {
        if ((label)w.size() != (label)lower_.size()) FatalErrorIn("lduAddressing::agglomeration", "face weights do not match the number of faces");
        if (hasProcessorPatches() || Pstream::parRun()) {
            if (!Pstream::parRun()) FatalErrorIn("GAMGAgglomeration::New", "processor patches outside a parallel run (Pstream::init)");
            std::vector<label> pr, pn;
            for (const lduInterface& i : interfaces_) { pr.push_back(i.neighbProcNo); pn.push_back(i.neighbPatchID); }
            miCheck(mi_gamg_create_coupled(handle(), w.data(), nCoarsest, mergeLevels, 1, Pstream::reduceComm(), Pstream::haloComm(),
                                           pr.data(), pn.data(), &gamg_), "GAMGAgglomeration::New");
        } else
        miCheck(mi_gamg_create(handle(), w.data(), nCoarsest, mergeLevels, 1, &gamg_), "GAMGAgglomeration::New");
        gamgCoarsest_ = nCoarsest; gamgMerge_ = mergeLevels;
    }
    return gamg_;
}

// ---- lduMatrix ----------------------------------------------------------------------------------------------
lduMatrix::lduMatrix(const lduAddressing& a)
: lduAddr_(a), diag_(a.size()), upper_((label)a.lowerAddrHost().size()), mat_(nullptr), dirty_(true) {}
lduMatrix::~lduMatrix() { if (mat_) mi_matrix_destroy(mat_); }
scalargpuField& lduMatrix::lower()
{
    if (!lowerPtr_) lowerPtr_.reset(new scalargpuField(upper_));
    dirty_ = true;
    return *lowerPtr_;
}
void lduMatrix::sync(const FieldFieldScalar* bou, const FieldFieldScalar* inte, const lduInterfaceFieldPtrsList* ifs) const
{
    if (!mat_) miCheck(mi_matrix_create(lduAddr_.handle(), &mat_), "lduMatrix::sync");
    if (dirty_) { // the reference's lowerSortPtr_ invalidation (lduMatrix.C:235,266)
        miCheck(mi_matrix_set_coeffs(mat_, diag_.data(), upper_.data(), lowerPtr_ ? lowerPtr_->data() : nullptr), "lduMatrix::sync");
        dirty_ = false;
    }
    const label nP = lduAddr_.nPatches();
    if (nP == 0 && Pstream::parRun() && !attached_) {
        miCheck(mi_matrix_attach_comm(mat_, Pstream::reduceComm(), Pstream::haloComm(), nullptr, nullptr,
                                      Pstream::returnReduceSum(lduAddr_.size())), "lduMatrix::sync");
        attached_ = true;
    }
    if (nP == 0) return;
    if (!bou || (label)bou->size() != nP || !ifs || (label)ifs->size() != nP)
        FatalErrorIn("lduMatrix::sync", "interface coefficient lists do not match the coupled patches of the addressing");
    const bool callerExt = !lduAddr_.engineCoupled();
    scalargpuField ext(callerExt ? mi_addr_n_ext(lduAddr_.handle()) : 0);
    label off = 0;
    for (label p = 0; p < nP; ++p) {
        const label n = (label)lduAddr_.patchAddr(p).size();
        miCheck(mi_matrix_set_interface_coeffs(mat_, p, (*bou)[p].data(), inte && (label)inte->size() == nP ? (*inte)[p].data() : nullptr), "lduMatrix::sync");
        if ((*ifs)[p]->doTransform() || transformSet_) { // transformCoupleField(pnf, cmpt) of this solve's component
            miCheck(mi_matrix_set_patch_transform(mat_, p, (*ifs)[p]->transformFactor(cmpt_)), "lduMatrix::sync");
            transformSet_ = true;
        }
        if (callerExt && n) miCopyD2D(ext.data() + off, (*ifs)[p]->patchNeighbourField.data(), sizeof(scalar) * n);
        off += n;
    }
    if (callerExt) miCheck(mi_matrix_set_ext(mat_, ext.data()), "lduMatrix::sync");
    if (!attached_ && (lduAddr_.hasProcessorPatches() || Pstream::parRun())) {
        // decomposed case: from here on the engine exchanges the processor-patch values and all-reduces the sums itself
        if (!Pstream::parRun()) FatalErrorIn("lduMatrix::sync", "processor patches outside a parallel run (Pstream::init)");
        std::vector<label> pr, pn;
        for (label p = 0; p < nP; ++p) { pr.push_back(lduAddr_.interface(p).neighbProcNo); pn.push_back(lduAddr_.interface(p).neighbPatchID); }
        miCheck(mi_matrix_attach_comm(mat_, Pstream::reduceComm(), Pstream::haloComm(), pr.data(), pn.data(),
                                      Pstream::returnReduceSum(lduAddr_.size())), "lduMatrix::sync");
        attached_ = true;
    }
    mi_ctx_synchronize(miEngine::New().ctx);
}
void lduMatrix::Amul(scalargpuField& Apsi, const scalargpuField& psi, const FieldFieldScalar& b, const lduInterfaceFieldPtrsList& ifs, direction cmpt) const
{
    cmpt_ = cmpt; sync(&b, nullptr, &ifs); miCheck(mi_amul(mat_, psi.data(), Apsi.data()), "lduMatrix::Amul");
}
void lduMatrix::Tmul(scalargpuField& Tpsi, const scalargpuField& psi, const FieldFieldScalar& i, const lduInterfaceFieldPtrsList& ifs, direction cmpt) const
{
    cmpt_ = cmpt;
    // Tmul uses interfaceIntCoeffs (lduMatrixATmul.C:264-342): pass them in both slots so the engine's lower side holds them
    sync(&i, &i, &ifs); miCheck(mi_tmul(mat_, psi.data(), Tpsi.data()), "lduMatrix::Tmul");
    dirty_ = true;
}
void lduMatrix::sumA(scalargpuField& s, const FieldFieldScalar& b, const lduInterfaceFieldPtrsList& ifs) const
{
    sync(&b, nullptr, &ifs); miCheck(mi_sumA(mat_, s.data()), "lduMatrix::sumA");
}
void lduMatrix::patchNeighbourField(FieldFieldScalar& nbr, const scalargpuField& psi, const FieldFieldScalar& b, const lduInterfaceFieldPtrsList& ifs) const
{
    sync(&b, nullptr, &ifs);
    label nExt = 0;
    for (const lduInterfaceField* f : ifs) nExt += (label)f->faceCells.size();
    scalargpuField all(nExt);
    miCheck(mi_matrix_patch_neighbour_field(mat_, psi.data(), all.data()), "coupledFvPatchField::patchNeighbourField");
    const std::vector<scalar> h = all.asHost();
    nbr.clear();
    std::size_t off = 0;
    for (const lduInterfaceField* f : ifs) {
        const std::size_t n = f->faceCells.size();
        nbr.push_back(scalargpuField(scalarField(h.begin() + (std::ptrdiff_t)off, h.begin() + (std::ptrdiff_t)(off + n))));
        off += n;
    }
}
void lduMatrix::residual(scalargpuField& rA, const scalargpuField& psi, const scalargpuField& source, const FieldFieldScalar& b,
                         const lduInterfaceFieldPtrsList& ifs, direction cmpt) const
{
    cmpt_ = cmpt; sync(&b, nullptr, &ifs); miCheck(mi_residual(mat_, psi.data(), source.data(), rA.data()), "lduMatrix::residual");
}
void lduMatrix::negSumDiag() { miCheck(mi_row_face_op(lduAddr_.handle(), 1, lowerPtr_ ? lowerPtr_->data() : nullptr, upper_.data(), diag_.data()), "lduMatrix::negSumDiag"); dirty_ = true; }
void lduMatrix::sumDiag() { miCheck(mi_row_face_op(lduAddr_.handle(), 0, lowerPtr_ ? lowerPtr_->data() : nullptr, upper_.data(), diag_.data()), "lduMatrix::sumDiag"); dirty_ = true; }
void lduMatrix::sumMagOffDiag(scalargpuField& s) const { miCheck(mi_row_face_op(lduAddr_.handle(), 2, lowerPtr_ ? lowerPtr_->data() : nullptr, upper_.data(), s.data()), "lduMatrix::sumMagOffDiag"); }

// ---- preconditioner names (lduMatrixPreconditioner.C:36-65) ---------------------------------------------------
word lduMatrix::preconditioner::getName(const dictionary& d)
{
    word name = d.lookup("preconditioner");
    // DIC / DILU (and friends) are replaced by the approximate inverse in the reference
    // (DICPreconditioner.C:42-58, DILUPreconditioner.C:42-58): the reported name becomes AINV
    if (name == "DIC" || name == "DILU") name = "AINV";
    return name;
}
int lduMatrix::preconditioner::kindFor(const dictionary& d, bool sym)
{
    const word name = d.lookup("preconditioner");
    const bool ok = name == "AINV" || name == "diagonal" || name == "none" || name == (sym ? "DIC" : "DILU");
    if (!ok)
        FatalErrorIn("lduMatrix::preconditioner::New(const solver&, const dictionary&)",
                     word("Unknown ") + (sym ? "symmetric" : "asymmetric") + " matrix preconditioner " + name + "\n\nValid " + (sym ? "symmetric" : "asymmetric")
                     + " matrix preconditioners :\n" + (sym ? "4(AINV DIC diagonal none)" : "4(AINV DILU diagonal none)"));
    return kind(getName(d));
}
int lduMatrix::preconditioner::kind(const word& n)
{
    if (n == "none") return MI_PRECOND_NONE;
    if (n == "diagonal") return MI_PRECOND_DIAGONAL;
    if (n == "AINV") return MI_PRECOND_AINV;
    FatalErrorIn("lduMatrix::preconditioner::New", "Unknown preconditioner " + n + "\n\nValid preconditioners are :\n(AINV DIC DILU diagonal none)");
}

// ---- solver base + factory --------------------------------------------------------------------------------------
std::map<word, lduMatrix::solver::ctor>& lduMatrix::solver::symMatrixConstructorTable() { static std::map<word, ctor> t; return t; }
std::map<word, lduMatrix::solver::ctor>& lduMatrix::solver::asymMatrixConstructorTable() { static std::map<word, ctor> t; return t; }

lduMatrix::solver::solver(const word& fieldName, const lduMatrix& matrix, const FieldFieldScalar& b, const FieldFieldScalar& i,
                          const lduInterfaceFieldPtrsList& ifs, const dictionary& d)
: fieldName_(fieldName), matrix_(matrix), interfaceBouCoeffs_(b), interfaceIntCoeffs_(i), interfaces_(ifs), controlDict_(d)
{
    readControls();
}
void lduMatrix::solver::readControls()
{
    maxIter_ = controlDict_.lookupOrDefault<label>("maxIter", 1000);
    minIter_ = controlDict_.lookupOrDefault<label>("minIter", 0);
    tolerance_ = controlDict_.lookupOrDefault<scalar>("tolerance", 1e-6);
    relTol_ = controlDict_.lookupOrDefault<scalar>("relTol", 0);
}

namespace
{
word tableNames(const std::map<word, lduMatrix::solver::ctor>& t)
{
    word s = "(";
    for (auto& kv : t) s += kv.first + " ";
    if (s.size() > 1) s.pop_back();
    return s + ")";
}
mi_solver_controls controlsOf(scalar tol, scalar relTol, label maxIter, label minIter) { mi_solver_controls c = {tol, relTol, maxIter, minIter}; return c; }
solverPerformance perfOf(const word& solverName, const word& fieldName, const mi_solver_perf& r)
{
    return solverPerformance(solverName, fieldName, r.initialResidual, r.finalResidual, r.nIterations, r.converged != 0, r.singular != 0);
}
// whole-solver calls own the exchange: fine for cyclic / processor interfaces (the engine does it), not for interfaces
// whose neighbour values the caller supplies once (they would go stale inside the iteration)
void requireUncoupled(const lduInterfaceFieldPtrsList& ifs, const char* who)
{
    for (const lduInterfaceField* f : ifs)
        if (f && f->type() == "coupled")
            FatalErrorIn(who, "interfaces with caller-supplied neighbour values cannot be iterated on: use cyclicLduInterfaceField / processorLduInterfaceField");
}
} // namespace

// diagonalSolver (solvers/diagonalSolver/diagonalSolver.C): psi = source/diag
class diagonalSolver : public lduMatrix::solver
{
public:
    using lduMatrix::solver::solver;
    solverPerformance solve(scalargpuField& psi, const scalargpuField& source, const direction) const override
    {
        std::vector<scalar> d = matrix_.diag().asHost(), s = source.asHost();
        for (std::size_t i = 0; i < d.size(); ++i) s[i] /= d[i];
        psi = s;
        return solverPerformance("diagonalSolver", fieldName_, 0, 0, 0, true, false);
    }
};

autoPtr<lduMatrix::solver> lduMatrix::solver::New(const word& fieldName, const lduMatrix& matrix, const FieldFieldScalar& b,
                                                  const FieldFieldScalar& i, const lduInterfaceFieldPtrsList& ifs, const dictionary& d)
{
    const word name = d.lookup("solver");
    if (matrix.diagonal()) return autoPtr<solver>(new diagonalSolver(fieldName, matrix, b, i, ifs, d));
    const bool sym = matrix.symmetric();
    auto& table = sym ? symMatrixConstructorTable() : asymMatrixConstructorTable();
    auto it = table.find(name);
    if (it == table.end())
        FatalErrorIn("lduMatrix::solver::New", "Unknown " + word(sym ? "symmetric" : "asymmetric") + " matrix solver " + name +
                     "\n\nValid " + word(sym ? "symmetric" : "asymmetric") + " matrix solvers are :\n" + tableNames(table));
    return it->second(fieldName, matrix, b, i, ifs, d);
}

// ---- concrete solvers, registered under the reference's run-time names ---------------------------------------------
class PCG : public lduMatrix::solver
{
public:
    using lduMatrix::solver::solver;
    solverPerformance solve(scalargpuField& psi, const scalargpuField& source, const direction cmpt) const override
    {
        requireUncoupled(interfaces_, "PCG::solve");
        const word pre = lduMatrix::preconditioner::getName(controlDict_);
        const mi_solver_controls c = controlsOf(tolerance_, relTol_, maxIter_, minIter_);
        mi_solver_perf r;
        miCheck(mi_pcg_solve(matrix_.handle(interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_, cmpt), psi.data(), source.data(), &c,
                             lduMatrix::preconditioner::kindFor(controlDict_, matrix_.symmetric()), &r, nullptr, 0), "PCG::solve");
        return perfOf(pre + "PCG", fieldName_, r); // PCG.C:75-80
    }
};
class PBiCG : public lduMatrix::solver
{
public:
    using lduMatrix::solver::solver;
    solverPerformance solve(scalargpuField& psi, const scalargpuField& source, const direction cmpt) const override
    {
        requireUncoupled(interfaces_, "PBiCG::solve");
        const word pre = lduMatrix::preconditioner::getName(controlDict_);
        const mi_solver_controls c = controlsOf(tolerance_, relTol_, maxIter_, minIter_);
        mi_solver_perf r;
        miCheck(mi_pbicg_solve(matrix_.handle(interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_, cmpt), psi.data(), source.data(), &c,
                               lduMatrix::preconditioner::kindFor(controlDict_, matrix_.symmetric()), &r, nullptr, 0), "PBiCG::solve");
        return perfOf(pre + "PBiCG", fieldName_, r);
    }
};
class PBiCGStab : public lduMatrix::solver
{
public:
    using lduMatrix::solver::solver;
    solverPerformance solve(scalargpuField& psi, const scalargpuField& source, const direction cmpt) const override
    {
        requireUncoupled(interfaces_, "PBiCGStab::solve");
        const word pre = lduMatrix::preconditioner::getName(controlDict_);
        const mi_solver_controls c = controlsOf(tolerance_, relTol_, maxIter_, minIter_);
        mi_solver_perf r;
        // keep the reference's `psi += omega*yA` (PBiCGStab.C:263-270) unless the case asks for the textbook update
        const int quirk = controlDict_.lookupOrDefault<label>("textbookOmegaUpdate", 0) ? 0 : 1;
        miCheck(mi_pbicgstab_solve(matrix_.handle(interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_, cmpt), psi.data(), source.data(), &c,
                                   lduMatrix::preconditioner::kindFor(controlDict_, matrix_.symmetric()), quirk, &r, nullptr, 0), "PBiCGStab::solve");
        return perfOf(pre + "PBiCGStab", fieldName_, r);
    }
};
class smoothSolver : public lduMatrix::solver
{
public:
    using lduMatrix::solver::solver;
    solverPerformance solve(scalargpuField& psi, const scalargpuField& source, const direction cmpt) const override
    {
        requireUncoupled(interfaces_, "smoothSolver::solve");
        const word sm = controlDict_.lookup("smoother");
        if (sm != "GaussSeidel" && sm != "Jacobi") // GaussSeidelSmoother.C:43-66: GaussSeidel IS Jacobi in the reference
            FatalErrorIn("lduMatrix::smoother::New", "Unknown smoother " + sm + "\n\nValid smoothers are :\n(GaussSeidel Jacobi)");
        const mi_solver_controls c = controlsOf(tolerance_, relTol_, maxIter_, minIter_);
        mi_solver_perf r;
        miCheck(mi_smooth_solve(matrix_.handle(interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_, cmpt), psi.data(), source.data(), &c,
                                controlDict_.lookupOrDefault<scalar>("omega", 0.9), controlDict_.lookupOrDefault<label>("nSweeps", 1),
                                &r, nullptr, 0), "smoothSolver::solve");
        return perfOf("smoothSolver", fieldName_, r);
    }
};
class GAMGSolver : public lduMatrix::solver
{
public:
    using lduMatrix::solver::solver;
    solverPerformance solve(scalargpuField& psi, const scalargpuField& source, const direction cmpt) const override
    {
        requireUncoupled(interfaces_, "GAMGSolver::solve");
        const word agg = controlDict_.lookupOrDefault<word>("agglomerator", "faceAreaPair");
        const label nCoarsest = controlDict_.lookupOrDefault<label>("nCellsInCoarsestLevel", -1);
        if (nCoarsest < 0) FatalErrorIn("GAMGAgglomeration::GAMGAgglomeration", "keyword nCellsInCoarsestLevel is undefined in dictionary"); // GAMGAgglomeration.C:96-99
        if (agg != "dummy" && !controlDict_.found("mergeLevels")) FatalErrorIn("pairGAMGAgglomeration::pairGAMGAgglomeration", "keyword mergeLevels is undefined in dictionary");
        if (agg == "dummy" && !controlDict_.found("nLevels")) FatalErrorIn("dummyAgglomeration::dummyAgglomeration", "keyword nLevels is undefined in dictionary"); // dummyAgglomeration.C:52
        {   // GAMGSolver.C:75: interpolateCorrection (default false).  The reference's interpolate() overload that every level
            // with a coarser level calls starts with notImplemented() (GAMGSolverInterpolate.C:180), i.e. it aborts: same here.
            const word ic = controlDict_.lookupOrDefault<word>("interpolateCorrection", "false");
            if (ic == "true" || ic == "on" || ic == "yes" || ic == "y" || ic == "t") FatalErrorIn("GAMGSolver::interpolate()", "Not implemented");
        }
        const label mergeLevels = controlDict_.lookupOrDefault<label>("mergeLevels", 1);
        if (mergeLevels < 1) FatalErrorIn("pairGAMGAgglomeration::pairGAMGAgglomeration", "mergeLevels must be positive");
        const word sm = controlDict_.lookupOrDefault<word>("smoother", "GaussSeidel");
        if (sm != "GaussSeidel" && sm != "Jacobi") FatalErrorIn("lduMatrix::smoother::New", "Unknown smoother " + sm);
        scalarField w;
        if (agg == "algebraicPair") { w = matrix_.upper().asHost(); for (scalar& v : w) v = std::fabs(v); } // algebraicPairGAMGAgglomeration.C:47-61
        else if (agg == "faceAreaPair") {
            const word key = "faceAreaPairWeights";  // supplied by the mesh layer (faceAreaPairGAMGAgglomeration.C:54-81)
            if (!faceWeights_) FatalErrorIn("faceAreaPairGAMGAgglomeration", "no face-area weights registered for this mesh (" + key + ")");
            w = *faceWeights_;
        } else if (agg != "dummy") FatalErrorIn("GAMGAgglomeration::New", "Unknown GAMGAgglomeration type " + agg);
        mi_gamg_t g = agg == "dummy" ? matrix_.lduAddr().dummyAgglomeration(controlDict_.lookupOrDefault<label>("nLevels", 1))
                                     : matrix_.lduAddr().agglomeration(w, nCoarsest, mergeLevels);
        mi_gamg_controls c;
        c.tolerance = tolerance_; c.relTol = relTol_; c.maxIter = maxIter_; c.minIter = minIter_;
        c.nPreSweeps = controlDict_.lookupOrDefault<label>("nPreSweeps", 0);
        c.preSweepsLevelMultiplier = controlDict_.lookupOrDefault<label>("preSweepsLevelMultiplier", 1);
        c.maxPreSweeps = controlDict_.lookupOrDefault<label>("maxPreSweeps", 4);
        c.nPostSweeps = controlDict_.lookupOrDefault<label>("nPostSweeps", 2);
        c.postSweepsLevelMultiplier = controlDict_.lookupOrDefault<label>("postSweepsLevelMultiplier", 1);
        c.maxPostSweeps = controlDict_.lookupOrDefault<label>("maxPostSweeps", 4);
        c.nFinestSweeps = controlDict_.lookupOrDefault<label>("nFinestSweeps", 2);
        c.scaleCorrection = controlDict_.found("scaleCorrection") ? controlDict_.lookupOrDefault<label>("scaleCorrection", 1) : -1;
        c.omega = controlDict_.lookupOrDefault<scalar>("omega", 0.9);
        {   // GAMGSolver.C:77,232: directSolveCoarsest (default true in this code base); false = ICCG / BICCG on the coarsest level
            const word ds = controlDict_.lookupOrDefault<word>("directSolveCoarsest", "true");
            c.directSolveCoarsest = (ds == "false" || ds == "off" || ds == "no" || ds == "n" || ds == "f") ? 0 : 1;
            c.reserved = 0;
        }
        mi_solver_perf r;
        miCheck(mi_gamg_solve(g, matrix_.handle(interfaceBouCoeffs_, interfaceIntCoeffs_, interfaces_, cmpt), psi.data(), source.data(), &c, &r, nullptr, 0), "GAMGSolver::solve");
        return perfOf("GAMG", fieldName_, r);
    }
    static const scalarField* faceWeights_;
};
const scalarField* GAMGSolver::faceWeights_ = nullptr;
void setFaceAreaPairWeights(const scalarField* w) { GAMGSolver::faceWeights_ = w; }

// static registration objects, as in PCG.C:36-37 etc.
static lduMatrix::solver::addsymMatrixConstructorToTable<PCG> addPCGSymMatrixConstructorToTable_("PCG");
static lduMatrix::solver::addasymMatrixConstructorToTable<PBiCG> addPBiCGAsymMatrixConstructorToTable_("PBiCG");
static lduMatrix::solver::addasymMatrixConstructorToTable<PBiCGStab> addPBiCGStabAsymMatrixConstructorToTable_("PBiCGStab");
// ICCG / BICCG (solvers/ICCG/ICCG.C:34-35, solvers/BICCG/BICCG.C:34-35): PCG / PBiCG under another run-time name; constructed
// from a dictionary they pass it on unchanged, so the preconditioner is the dictionary's and the printed name is PCG's / PBiCG's
class ICCG : public PCG { public: using PCG::PCG; };
class BICCG : public PBiCG { public: using PBiCG::PBiCG; };
static lduMatrix::solver::addsymMatrixConstructorToTable<ICCG> addICCGSymMatrixConstructorToTable_("ICCG");
static lduMatrix::solver::addasymMatrixConstructorToTable<BICCG> addBICCGSymMatrixConstructorToTable_("BICCG");
static lduMatrix::solver::addsymMatrixConstructorToTable<smoothSolver> addsmoothSolverSymMatrixConstructorToTable_("smoothSolver");
static lduMatrix::solver::addasymMatrixConstructorToTable<smoothSolver> addsmoothSolverAsymMatrixConstructorToTable_("smoothSolver");
static lduMatrix::solver::addsymMatrixConstructorToTable<GAMGSolver> addGAMGSolverMatrixConstructorToTable_("GAMG");
static lduMatrix::solver::addasymMatrixConstructorToTable<GAMGSolver> addGAMGAsymSolverMatrixConstructorToTable_("GAMG");

// ---- fvScalarMatrix -----------------------------------------------------------------------------------------------------
fvScalarMatrix::fvScalarMatrix(const word& psiName, const lduAddressing& a, const std::vector<labelList>& pfc, const std::vector<bool>& coupled)
: lduMatrix(a), psiName_(psiName), source_(a.size()), patchFaceCells_(pfc), patchCoupled_(coupled)
{
    for (const labelList& p : pfc) { internalCoeffs_.emplace_back((label)p.size()); boundaryCoeffs_.emplace_back((label)p.size()); }
    patches_.assign(pfc.size(), nullptr);
}
fvScalarMatrix::~fvScalarMatrix() { for (mi_patch_t p : patches_) if (p) mi_patch_destroy(p); }
static mi_patch_t patchOf(std::vector<mi_patch_t>& cache, std::size_t k, label nCells, const labelList& fc)
{
    if (!cache[k]) miCheck(mi_patch_create(miEngine::New().ctx, nCells, (label)fc.size(), fc.data(), &cache[k]), "fvPatch::faceCells");
    return cache[k];
}
void fvScalarMatrix::addBoundaryDiag(scalargpuField& diag) const
{
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p)
        miCheck(mi_patch_add(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]), internalCoeffs_[p].data(), diag.data(), 0), "fvMatrix::addBoundaryDiag");
}
void fvScalarMatrix::addBoundarySource(scalargpuField& source, const std::vector<const scalargpuField*>& patchNeighbourField) const
{
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p) {
        if (!patchCoupled_[p])
            miCheck(mi_patch_add(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]), boundaryCoeffs_[p].data(), source.data(), 0), "fvMatrix::addBoundarySource");
        else if (p < patchNeighbourField.size() && patchNeighbourField[p])
            miCheck(mi_patch_add_product(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]), boundaryCoeffs_[p].data(),
                                         patchNeighbourField[p]->data(), source.data(), 0), "fvMatrix::addBoundarySource");
    }
}
void fvScalarMatrix::relax(scalar alpha, const scalargpuField& psi)
{
    std::vector<mi_patch_t> ph; std::vector<const double*> ic, bc; std::vector<int32_t> cp;
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p) {
        ph.push_back(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]));
        ic.push_back(internalCoeffs_[p].data()); bc.push_back(boundaryCoeffs_[p].data()); cp.push_back(patchCoupled_[p] ? 1 : 0);
    }
    miCheck(mi_relax(lduAddr().handle(), alpha, diag().data(), asymmetric() ? lower().data() : nullptr, upper().data(), source_.data(), psi.data(),
                     (label)ph.size(), ph.data(), ic.data(), bc.data(), cp.data()), "fvMatrix::relax");
}
void fvScalarMatrix::setReference(label celli, scalar value)
{
    miCheck(mi_fvm_set_reference(lduAddr().handle(), celli, value, diag().data(), source_.data()), "fvMatrix::setReference");
}
void fvScalarMatrix::setValues(const labelgpuList& cellLabels, const scalargpuField& values, scalargpuField& psi)
{
    std::vector<mi_patch_t> ph; std::vector<double*> ic, bc;
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p) {
        ph.push_back(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]));
        ic.push_back(internalCoeffs_[p].data()); bc.push_back(boundaryCoeffs_[p].data());
    }
    const bool wasSym = !asymmetric();
    const scalargpuField upperIn(upper());            // (the non-const lower() below copies upper, as the reference's does)
    scalargpuField& lo = lower();
    miCheck(mi_fvm_set_values(lduAddr().handle(), cellLabels.size(), cellLabels.data(), values.data(), 0, psi.data(), diag().data(), source_.data(),
                              upperIn.data(), wasSym ? nullptr : lo.data(), upper().data(), lo.data(), (label)ph.size(), ph.data(), ic.data(), bc.data()),
            "fvMatrix::setValues");
}
solverPerformance fvScalarMatrix::solve(scalargpuField& psi, const dictionary& solverControls)
{
    // fvScalarMatrix.C:142-192: saveDiag; addBoundaryDiag; totalSource = source + boundary; solver::New()->solve; restore
    scalargpuField saveDiag(diag());
    addBoundaryDiag(diag());
    scalargpuField totalSource(source_);
    addBoundarySource(totalSource);
    FieldFieldScalar noCoeffs; lduInterfaceFieldPtrsList noInterfaces;
    solverPerformance perf = lduMatrix::solver::New(psiName_, *this, noCoeffs, noCoeffs, noInterfaces, solverControls)->solve(psi, totalSource);
    perf.print(Info);
    diag() = saveDiag;
    return perf;
}

solverPerformance max(const solverPerformance& a, const solverPerformance& b)
{
    solverPerformance r(a.solverName(), a.fieldName(), std::max(a.initialResidual(), b.initialResidual()),
                        std::max(a.finalResidual(), b.finalResidual()), std::max(a.nIterations(), b.nIterations()),
                        a.converged() && b.converged(), a.singular() || b.singular());
    return r;
}

// ---- fvVectorMatrix ---------------------------------------------------------------------------------------------------
fvVectorMatrix::fvVectorMatrix(const word& psiName, const lduAddressing& a, const std::vector<labelList>& pfc, const std::vector<bool>& coupled)
: lduMatrix(a), psiName_(psiName), source_(a.size()), patchFaceCells_(pfc), patchCoupled_(coupled)
{
    for (const labelList& p : pfc) { internalCoeffs_.emplace_back((label)p.size()); boundaryCoeffs_.emplace_back((label)p.size()); }
    patches_.assign(pfc.size(), nullptr);
}
fvVectorMatrix::~fvVectorMatrix() { for (mi_patch_t p : patches_) if (p) mi_patch_destroy(p); }
void fvVectorMatrix::addBoundaryDiag(scalargpuField& diag, direction cmpt) const
{
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p)
        miCheck(mi_patch_add(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]), internalCoeffs_[p].component(cmpt).data(), diag.data(), 0), "fvMatrix::addBoundaryDiag");
}
void fvVectorMatrix::addBoundarySource(vectorgpuField& source) const
{
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p)
        if (!patchCoupled_[p])
            for (direction d = 0; d < 3; ++d)
                miCheck(mi_patch_add(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]), boundaryCoeffs_[p].component(d).data(), source.component(d).data(), 0), "fvMatrix::addBoundarySource");
}
solverPerformance fvVectorMatrix::solve(vectorgpuField& psi, const dictionary& solverControls)
{
    static const char* componentNames[3] = {"x", "y", "z"};
    solverPerformance solverPerfVec("fvMatrix<Type>::solveSegregated", psiName_);
    scalargpuField saveDiag(diag());
    vectorgpuField source(lduAddr().size());
    for (direction d = 0; d < 3; ++d) source.component(d) = source_.component(d);
    addBoundarySource(source);
    FieldFieldScalar noCoeffs; lduInterfaceFieldPtrsList noInterfaces;
    // MI355X form of the component loop: PBiCG on the three components as ONE solve (mi_pbicg_solve_multi) -- every pass over
    // upper / lower serves all components, each component keeps its own diagonal (addBoundaryDiag(diag, cmpt)), scalars,
    // convergence test and solverPerformance line; same numbers as the loop below (`segregatedLoop yes;` selects that)
    if (asymmetric() && solverControls.lookup("solver") == "PBiCG" && solverControls.lookupOrDefault<word>("segregatedLoop", "no") != "yes") {
        std::vector<scalargpuField> dc;
        dc.reserve(3);
        const double* dptr[3]; double* pptr[3]; const double* sptr[3];
        for (direction cmpt = 0; cmpt < 3; ++cmpt) {
            dc.emplace_back(saveDiag);
            addBoundaryDiag(dc.back(), cmpt);
        }
        for (direction cmpt = 0; cmpt < 3; ++cmpt) { dptr[cmpt] = dc[cmpt].data(); pptr[cmpt] = psi.component(cmpt).data(); sptr[cmpt] = source.component(cmpt).data(); }
        const mi_solver_controls c = controlsOf(solverControls.lookupOrDefault<scalar>("tolerance", 1e-6), solverControls.lookupOrDefault<scalar>("relTol", 0),
                                                solverControls.lookupOrDefault<label>("maxIter", 1000), solverControls.lookupOrDefault<label>("minIter", 0));
        mi_solver_perf r[3];
        miCheck(mi_pbicg_solve_multi(handle(noCoeffs, noCoeffs, noInterfaces, 0), 3, dptr, pptr, sptr, &c,
                                     lduMatrix::preconditioner::kindFor(solverControls, false), r, nullptr, 0), "fvMatrix<Type>::solveSegregated");
        const word pre = lduMatrix::preconditioner::getName(solverControls);
        for (direction cmpt = 0; cmpt < 3; ++cmpt) {
            solverPerformance solverPerf = perfOf(pre + "PBiCG", psiName_ + componentNames[cmpt], r[cmpt]);
            solverPerf.print(Info);
            solverPerformance m = max(solverPerfVec, solverPerf);
            solverPerfVec = solverPerformance(solverPerf.solverName(), psiName_, m.initialResidual(), m.finalResidual(), m.nIterations(), m.converged(), m.singular());
        }
        return solverPerfVec;
    }
    for (direction cmpt = 0; cmpt < 3; ++cmpt) {
        addBoundaryDiag(diag(), cmpt);
        solverPerformance solverPerf = lduMatrix::solver::New(psiName_ + componentNames[cmpt], *this, noCoeffs, noCoeffs, noInterfaces, solverControls)
                                           ->solve(psi.component(cmpt), source.component(cmpt), cmpt);
        solverPerf.print(Info);
        solverPerformance m = max(solverPerfVec, solverPerf);
        solverPerfVec = solverPerformance(solverPerf.solverName(), psiName_, m.initialResidual(), m.finalResidual(), m.nIterations(),
                                          m.converged(), m.singular()); // (the reference's vector perf starts unconverged, so it stays so)
        diag() = saveDiag;
    }
    return solverPerfVec;
}

void fvVectorMatrix::A(scalargpuField& Aphi, const scalargpuField& V) const
{
    mi_ctx_t ctx = miEngine::New().ctx;
    Aphi = diag();
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p) {
        const label np = (label)patchFaceCells_[p].size();
        if (np == 0) continue;
        scalargpuField av(np), three(std::vector<scalar>((std::size_t)np, 3.0));       // cmptAv(v) = (v.x() + v.y() + v.z())/3
        miCheck(mi_vec_axpby(ctx, np, 1.0, internalCoeffs_[p].component(0).data(), 1.0, internalCoeffs_[p].component(1).data(), av.data()), "cmptAv");
        miCheck(mi_vec_axpby(ctx, np, 1.0, av.data(), 1.0, internalCoeffs_[p].component(2).data(), av.data()), "cmptAv");
        miCheck(mi_vec_div(ctx, np, av.data(), three.data(), av.data()), "cmptAv");
        miCheck(mi_patch_add(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]), av.data(), Aphi.data(), 0), "fvMatrix::addCmptAvBoundaryDiag");
    }
    miCheck(mi_vec_div(ctx, Aphi.size(), Aphi.data(), V.data(), Aphi.data()), "fvMatrix::A");
}
void fvVectorMatrix::H(vectorgpuField& Hphi, const vectorgpuField& psi, const scalargpuField& V) const
{
    static const FieldFieldScalar none; static const lduInterfaceFieldPtrsList noIfs;
    mi_ctx_t ctx = miEngine::New().ctx;
    for (direction d = 0; d < 3; ++d) {
        miCheck(mi_H(handle(none, none, noIfs), psi.component(d).data(), Hphi.component(d).data()), "lduMatrix::H");
        miCheck(mi_vec_axpby(ctx, Hphi.size(), 1.0, Hphi.component(d).data(), 1.0, source_.component(d).data(), Hphi.component(d).data()), "fvMatrix::H");
    }
    addBoundarySource(Hphi);
    for (direction d = 0; d < 3; ++d) miCheck(mi_vec_div(ctx, Hphi.size(), Hphi.component(d).data(), V.data(), Hphi.component(d).data()), "fvMatrix::H");
}
void fvm::assemble(fvVectorMatrix& M, scalar rDeltaT, scalar rho, const scalargpuField& V, const vectorgpuField& psiOld, const scalargpuField* faceFlux,
                   const scalargpuField* weights, const scalargpuField* deltaCoeffs, const scalargpuField* gammaMagSf)
{
    mi_fvm_terms t{};
    t.ddt = 1; t.r_delta_t = rDeltaT; t.rho_value = rho; t.vol_dev = V.data();
    t.div_flux_dev = faceFlux ? faceFlux->data() : nullptr; t.div_weights_dev = weights ? weights->data() : nullptr;
    t.lap_delta_coeffs_dev = deltaCoeffs ? deltaCoeffs->data() : nullptr; t.lap_gamma_magsf_dev = gammaMagSf ? gammaMagSf->data() : nullptr;
    const double* po[3] = {psiOld.component(0).data(), psiOld.component(1).data(), psiOld.component(2).data()};
    t.n_rhs = 3; t.psi_old_dev = po;
    double* so[3] = {M.source().component(0).data(), M.source().component(1).data(), M.source().component(2).data()};
    miCheck(mi_fvm_assemble(M.lduAddr().handle(), &t, faceFlux ? M.lower().data() : nullptr, M.upper().data(), M.diag().data(), so, nullptr), "fvm::assemble");
}
void fvm::laplacian(fvScalarMatrix& M, const scalargpuField& deltaCoeffs, const scalargpuField& gammaMagSf)
{
    miCheck(mi_fvm_laplacian(M.lduAddr().handle(), deltaCoeffs.data(), gammaMagSf.data(), M.upper().data(), M.diag().data()), "fvm::laplacian");
}
void fvm::div(fvScalarMatrix& M, const scalargpuField& weights, const scalargpuField& faceFlux)
{
    scalargpuField& lower = M.lower();
    miCheck(mi_fvm_div(M.lduAddr().handle(), weights.data(), faceFlux.data(), lower.data(), M.upper().data(), M.diag().data()), "fvm::div");
}

// ---- fvMatrix operators and the scheme front-end ------------------------------------------------------------------
void fvScalarMatrix::axpyFrom(const fvScalarMatrix& B, scalar b)
{
    if (&B.lduAddr() != &lduAddr()) FatalErrorIn("fvMatrix::operator+=", "incompatible matrices: different addressing");
    mi_ctx_t ctx = miEngine::New().ctx;
    auto ax = [&](scalargpuField& x, const scalargpuField& y) {
        if (x.size() != y.size()) FatalErrorIn("fvMatrix::operator+=", "incompatible fields");
        if (x.size()) miCheck(mi_vec_axpby(ctx, x.size(), 1.0, x.data(), b, y.data(), x.data()), "fvMatrix::operator+=");
    };
    if (B.asymmetric() && symmetric()) lower();                 // promote: lower becomes a copy of upper first
    if (asymmetric()) ax(lower(), B.lower());                    // B.lower() aliases B.upper() when B is symmetric
    ax(upper(), B.upper()); ax(diag(), B.diag()); ax(source_, B.source_);
    for (std::size_t p = 0; p < internalCoeffs_.size(); ++p) { ax(internalCoeffs_[p], B.internalCoeffs_[p]); ax(boundaryCoeffs_[p], B.boundaryCoeffs_[p]); }
}
void fvScalarMatrix::A(scalargpuField& Aphi, const scalargpuField& V) const
{
    Aphi = diag();
    addBoundaryDiag(Aphi);
    miCheck(mi_vec_div(miEngine::New().ctx, Aphi.size(), Aphi.data(), V.data(), Aphi.data()), "fvMatrix::A");
}
void fvScalarMatrix::H(scalargpuField& Hphi, const scalargpuField& psi, const scalargpuField& V,
                       const std::vector<const scalargpuField*>& patchNeighbourField) const
{
    static const FieldFieldScalar none; static const lduInterfaceFieldPtrsList noIfs;
    miCheck(mi_H(handle(none, none, noIfs), psi.data(), Hphi.data()), "lduMatrix::H");
    mi_ctx_t ctx = miEngine::New().ctx;
    miCheck(mi_vec_axpby(ctx, Hphi.size(), 1.0, Hphi.data(), 1.0, source_.data(), Hphi.data()), "fvMatrix::H");
    addBoundarySource(Hphi, patchNeighbourField);
    miCheck(mi_vec_div(ctx, Hphi.size(), Hphi.data(), V.data(), Hphi.data()), "fvMatrix::H");
}
void fvScalarMatrix::flux(scalargpuField& internalFlux, FieldFieldScalar& boundaryFlux, const scalargpuField& psi,
                          const std::vector<const scalargpuField*>& patchNeighbourField) const
{
    static const FieldFieldScalar none; static const lduInterfaceFieldPtrsList noIfs;
    miCheck(mi_faceH(handle(none, none, noIfs), psi.data(), internalFlux.data()), "lduMatrix::faceH");
    if (boundaryFlux.size() != patchFaceCells_.size()) boundaryFlux.resize(patchFaceCells_.size());
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p) {
        if (boundaryFlux[p].size() != (label)patchFaceCells_[p].size()) boundaryFlux[p] = scalargpuField((label)patchFaceCells_[p].size());
        const scalargpuField* nbr = (patchCoupled_[p] && p < patchNeighbourField.size()) ? patchNeighbourField[p] : nullptr;
        if (patchCoupled_[p] && !nbr) throw error("fvMatrix::flux: coupled patch without patchNeighbourField");
        miCheck(mi_patch_flux(patchOf(patches_, p, lduAddr().size(), patchFaceCells_[p]), internalCoeffs_[p].data(), boundaryCoeffs_[p].data(),
                              psi.data(), nbr ? nbr->data() : nullptr, boundaryFlux[p].data()), "fvMatrix::flux");
    }
}
void fvScalarMatrix::nonOrthCorrection(const scalargpuField& vf, const std::vector<const scalargpuField*>& patchValues, const vectorgpuField& Sf,
                                       const std::vector<const vectorgpuField*>& patchSf, const scalargpuField& weights, const vectorgpuField& corrVecs,
                                       const scalargpuField& gammaMagSf, const scalargpuField& V)
{
    const lduAddressing& a = lduAddr();
    const label n = a.size(), nI = weights.size();
    mi_ctx_t ctx = miEngine::New().ctx;
    scalargpuField ssf(nI);
    miCheck(mi_face_interpolate(a.handle(), weights.data(), vf.data(), ssf.data()), "linear::interpolate");
    vectorgpuField g(n);                                                          // gaussGrad::gradf: internal faces, then every patch
    miCheck(mi_gauss_grad(a.handle(), Sf.component(0).data(), Sf.component(1).data(), Sf.component(2).data(), ssf.data(), nullptr,
                          g.component(0).data(), g.component(1).data(), g.component(2).data()), "gaussGrad::gradf");
    for (std::size_t p = 0; p < patchFaceCells_.size(); ++p) {
        const label np = (label)patchFaceCells_[p].size();
        if (np == 0) continue;
        mi_patch_t P = patchOf(patches_, p, n, patchFaceCells_[p]);
        scalargpuField pif(np);
        const scalargpuField* pv = p < patchValues.size() ? patchValues[p] : nullptr;
        if (!pv) { miCheck(mi_patch_internal_field(P, vf.data(), pif.data()), "fvPatchField::patchInternalField"); pv = &pif; }
        for (direction d = 0; d < 3; ++d)
            miCheck(mi_patch_add_product(P, patchSf[p]->component(d).data(), pv->data(), g.component(d).data(), 0), "gaussGrad::gradf (patch)");
    }
    for (direction d = 0; d < 3; ++d) miCheck(mi_vec_div(ctx, n, g.component(d).data(), V.data(), g.component(d).data()), "gaussGrad /= V");
    scalargpuField flux(nI), div(n);
    miCheck(mi_sngrad_correction_flux(a.handle(), corrVecs.component(0).data(), corrVecs.component(1).data(), corrVecs.component(2).data(), weights.data(),
                                      g.component(0).data(), g.component(1).data(), g.component(2).data(), gammaMagSf.data(), flux.data()),
            "correctedSnGrad::correction");
    miCheck(mi_surface_integrate(a.handle(), flux.data(), V.data(), div.data()), "fvc::div");
    miCheck(mi_vec_submul(ctx, n, V.data(), div.data(), source_.data()), "fvm::laplacian: source -= V*div(correction)");
}
fvScalarMatrix& fvScalarMatrix::operator+=(const fvScalarMatrix& B) { axpyFrom(B, 1.0); return *this; }
fvScalarMatrix& fvScalarMatrix::operator-=(const fvScalarMatrix& B) { axpyFrom(B, -1.0); return *this; }
fvScalarMatrix& fvScalarMatrix::operator*=(scalar s)
{
    mi_ctx_t ctx = miEngine::New().ctx;
    auto sc = [&](scalargpuField& x) { if (x.size()) miCheck(mi_vec_axpby(ctx, x.size(), s, x.data(), 0.0, x.data(), x.data()), "fvMatrix::operator*="); };
    if (asymmetric()) sc(lower());
    sc(upper()); sc(diag()); sc(source_);
    for (std::size_t p = 0; p < internalCoeffs_.size(); ++p) { sc(internalCoeffs_[p]); sc(boundaryCoeffs_[p]); }
    return *this;
}
void fvm::ddt(fvScalarMatrix& M, scalar rDeltaT, scalar rho, const scalargpuField& V, const scalargpuField& psiOld)
{
    miCheck(mi_fvm_ddt_euler(miEngine::New().ctx, V.size(), rDeltaT, rho, V.data(), psiOld.data(), M.diag().data(), M.source().data()), "fvm::ddt");
}
void fvm::assemble(fvScalarMatrix& M, scalar rDeltaT, scalar rho, const scalargpuField& V, const scalargpuField& psiOld, const scalargpuField* faceFlux,
                   const scalargpuField* weights, const scalargpuField* deltaCoeffs, const scalargpuField* gammaMagSf, const scalargpuField* su)
{
    mi_fvm_terms t{};
    t.ddt = 1; t.r_delta_t = rDeltaT; t.rho_value = rho; t.vol_dev = V.data();
    t.div_flux_dev = faceFlux ? faceFlux->data() : nullptr; t.div_weights_dev = weights ? weights->data() : nullptr;
    t.lap_delta_coeffs_dev = deltaCoeffs ? deltaCoeffs->data() : nullptr; t.lap_gamma_magsf_dev = gammaMagSf ? gammaMagSf->data() : nullptr;
    const double* po[1] = {psiOld.data()}; t.n_rhs = 1; t.psi_old_dev = po;
    const double* sd[1] = {su ? su->data() : nullptr}; const double sg[1] = {1.0};
    if (su) { t.n_su = 1; t.su_dev = sd; t.su_sign = sg; }
    double* so[1] = {M.source().data()};
    miCheck(mi_fvm_assemble(M.lduAddr().handle(), &t, faceFlux ? M.lower().data() : nullptr, M.upper().data(), M.diag().data(), so, nullptr), "fvm::assemble");
}
void fvc::grad(vectorgpuField& g, const lduAddressing& a, const vectorgpuField& Sf, const scalargpuField& ssf, const scalargpuField& V)
{
    miCheck(mi_gauss_grad(a.handle(), Sf.component(0).data(), Sf.component(1).data(), Sf.component(2).data(), ssf.data(), V.data(),
                          g.component(0).data(), g.component(1).data(), g.component(2).data()), "fvc::grad");
}
void fvc::interpolate(scalargpuField& sf, const lduAddressing& a, const scalargpuField& weights, const scalargpuField& vf)
{
    miCheck(mi_face_interpolate(a.handle(), weights.data(), vf.data(), sf.data()), "fvc::interpolate");
}
void fvc::surfaceIntegrate(scalargpuField& ivf, const lduAddressing& a, const scalargpuField& ssf, const scalargpuField* V)
{
    miCheck(mi_surface_integrate(a.handle(), ssf.data(), V ? V->data() : nullptr, ivf.data()), "fvc::surfaceIntegrate");
}
void fvc::fluxDiv(scalargpuField& phi, scalargpuField* divOut, const lduAddressing& a, const scalargpuField& weights, const vectorgpuField& Sf,
                  const vectorgpuField& Vf, const scalargpuField* addA, const scalargpuField* addB, const scalargpuField* V)
{
    scalargpuField scratch(divOut ? 0 : a.size());
    miCheck(mi_flux_div(a.handle(), weights.data(), Sf.component(0).data(), Sf.component(1).data(), Sf.component(2).data(),
                        Vf.component(0).data(), Vf.component(1).data(), Vf.component(2).data(), nullptr, addA ? addA->data() : nullptr,
                        addB ? addB->data() : nullptr, phi.data(), V ? V->data() : nullptr, divOut ? divOut->data() : scratch.data()), "fvc::flux + fvc::div");
}
void fvc::snGradCorrectionFlux(scalargpuField& flux, const lduAddressing& a, const vectorgpuField& corrVecs, const scalargpuField& weights,
                               const vectorgpuField& gradVf, const scalargpuField& gammaMagSf)
{
    miCheck(mi_sngrad_correction_flux(a.handle(), corrVecs.component(0).data(), corrVecs.component(1).data(), corrVecs.component(2).data(), weights.data(),
                                      gradVf.component(0).data(), gradVf.component(1).data(), gradVf.component(2).data(), gammaMagSf.data(), flux.data()),
            "correctedSnGrad::correction");
}
void fvc::ddtCorr(scalargpuField& out, const lduAddressing& a, scalar rDeltaT, const scalargpuField& weights, const vectorgpuField& Sf,
                  const vectorgpuField& Uold, const scalargpuField& phiOld)
{
    miCheck(mi_ddt_phi_corr(a.handle(), rDeltaT, weights.data(), Sf.component(0).data(), Sf.component(1).data(), Sf.component(2).data(),
                            Uold.component(0).data(), Uold.component(1).data(), Uold.component(2).data(), nullptr, phiOld.data(), out.data()), "fvc::ddtCorr");
}
void fieldAxpby(scalargpuField& out, scalar a, const scalargpuField& x, scalar b, const scalargpuField& y)
{
    miCheck(mi_vec_axpby(miEngine::New().ctx, out.size(), a, x.data(), b, y.data(), out.data()), "gpuField: a x + b y");
}
void fieldDivide(scalargpuField& out, const scalargpuField& x, const scalargpuField& y)
{
    miCheck(mi_vec_div(miEngine::New().ctx, out.size(), x.data(), y.data(), out.data()), "gpuField: x / y");
}
void fieldSubMul(scalargpuField& inout, const scalargpuField& x, const scalargpuField& y)
{
    miCheck(mi_vec_submul(miEngine::New().ctx, inout.size(), x.data(), y.data(), inout.data()), "gpuField: -= x*y");
}
fvPatchCells::fvPatchCells(label nCells, const labelList& faceCells) : h_(nullptr), n_((label)faceCells.size())
{
    miCheck(mi_patch_create(miEngine::New().ctx, nCells, n_, faceCells.data(), &h_), "fvPatch::faceCells");
}
fvPatchCells::~fvPatchCells() { if (h_) mi_patch_destroy(h_); }
void fvPatchCells::add(const scalargpuField& pf, scalargpuField& intf, bool subtract) const
{
    if (n_) miCheck(mi_patch_add(h_, pf.data(), intf.data(), subtract ? 1 : 0), "fvPatch: cells += patch field");
}
void fvPatchCells::addProduct(const scalargpuField& pf, const scalargpuField& q, scalargpuField& intf, bool subtract) const
{
    if (n_) miCheck(mi_patch_add_product(h_, pf.data(), q.data(), intf.data(), subtract ? 1 : 0), "fvPatch: cells += patch field * patch field");
}
void fvPatchCells::patchInternalField(const scalargpuField& psi, scalargpuField& out) const
{
    if (n_) miCheck(mi_patch_internal_field(h_, psi.data(), out.data()), "fvPatchField::patchInternalField");
}
void upwindWeights(scalargpuField& w, const scalargpuField& faceFlux)
{
    miCheck(mi_upwind_weights(miEngine::New().ctx, faceFlux.size(), faceFlux.data(), w.data()), "upwind::weights");
}
void limitedLinearWeights(scalargpuField& w, const lduAddressing& a, scalar k, const scalargpuField& cdWeights, const scalargpuField& faceFlux,
                          const scalargpuField& vf, const vectorgpuField& gradVf, const vectorgpuField& C)
{
    miCheck(mi_limited_linear_weights(a.handle(), k, cdWeights.data(), faceFlux.data(), vf.data(), gradVf.component(0).data(),
                                      gradVf.component(1).data(), gradVf.component(2).data(), C.component(0).data(), C.component(1).data(),
                                      C.component(2).data(), w.data(), nullptr), "limitedLinear::weights");
}

// ---- Pstream: the parallel run as this path sees it -------------------------------------------------------------
namespace {
struct PstreamState { bool par = false; int rank = 0, n = 1; mi_comm_t red = nullptr, halo = nullptr; };
PstreamState& pstream() { static PstreamState s; return s; }
}
bool Pstream::parRun() { return pstream().par; }
int Pstream::myProcNo() { return pstream().rank; }
int Pstream::nProcs() { return pstream().n; }
mi_comm_t Pstream::reduceComm() { return pstream().red; }
mi_comm_t Pstream::haloComm() { return pstream().halo; }
void Pstream::init(int myProcNo, int nProcs, const std::string& idFile)
{
    PstreamState& P = pstream();
    if (P.par) FatalErrorIn("Pstream::init", "already initialised");
    if (nProcs < 1 || myProcNo < 0 || myProcNo >= nProcs) FatalErrorIn("Pstream::init", "bad rank / size");
    unsigned char ids[256];
    if (myProcNo == 0) {
        miCheck(mi_comm_unique_id(ids, 128), "Pstream::init");
        miCheck(mi_comm_unique_id(ids + 128, 128), "Pstream::init");
        if (nProcs > 1) { // publish atomically: write beside, then rename
            const std::string tmp = idFile + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(ids, 1, 256, f) != 256) FatalErrorIn("Pstream::init", "cannot write " + tmp);
            fclose(f);
            if (rename(tmp.c_str(), idFile.c_str()) != 0) FatalErrorIn("Pstream::init", "cannot publish " + idFile);
        }
    } else {
        bool got = false;
        for (int attempt = 0; attempt < 6000 && !got; ++attempt) { // up to ~10 min
            FILE* f = fopen(idFile.c_str(), "rb");
            if (f) { got = fread(ids, 1, 256, f) == 256; fclose(f); }
            if (!got) { struct timespec ts = {0, 100 * 1000 * 1000}; nanosleep(&ts, nullptr); }
        }
        if (!got) FatalErrorIn("Pstream::init", "rank 0 never published the communicator ids in " + idFile);
    }
    mi_ctx_t ctx = miEngine::New().ctx;
    miCheck(mi_comm_create(ctx, nProcs, myProcNo, ids, &P.red), "Pstream::init");
    miCheck(mi_comm_create(ctx, nProcs, myProcNo, ids + 128, &P.halo), "Pstream::init");
    P.par = true; P.rank = myProcNo; P.n = nProcs;
}
void Pstream::exit()
{
    PstreamState& P = pstream();
    if (P.halo) mi_comm_destroy(P.halo);
    if (P.red) mi_comm_destroy(P.red);
    P = PstreamState();
}
scalar Pstream::returnReduceSum(scalar v)
{
    if (!parRun()) return v;
    scalargpuField t(1);
    t = scalarField(1, v);
    miCheck(mi_comm_allreduce_sum(reduceComm(), t.data(), 1), "returnReduce");
    return t.asHost()[0];
}
label Pstream::returnReduceSum(label v)
{
    if (!parRun()) return v;
    scalargpuField t(1);
    t = scalarField(1, (scalar)v);
    miCheck(mi_comm_allreduce_sum(reduceComm(), t.data(), 1), "returnReduce");
    return (label)(t.asHost()[0] + 0.5);
}

} // namespace Foam
