// Translation unit that compiles the REFERENCE's PCG.C, PBiCG.C, PBiCGStab.C and smoothSolver.C where they lie, for the GPU,
// on top of the engine's C ABI (see foam_engine_shim.H).  REF_LDU = ".../src/OpenFOAM/matrices/lduMatrix"
#include "foam_engine_shim.H"
#define PCG_H
#define PBiCG_H
#define PBiCGStab_H
#define PCGCache_H
#define smoothSolver_H
#define lduMatrix_H
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
namespace Foam { refContext ctx = {0, 0, 1, 0.9, 0, false}; int lduMatrix::debug = 0;
const scalar solverPerformance::great_ = 1e20; const scalar solverPerformance::small_ = 1e-20; const scalar solverPerformance::vsmall_ = 1e-300; }
#include REF_FILE(lduMatrix/lduMatrixSolverFunctors.H)
#include REF_FILE(lduMatrix/lduMatrixFunctors.H)
#include REF_FILE(solvers/PCG/PCG.C)
#include REF_FILE(solvers/PBiCG/PBiCG.C)
#include REF_FILE(solvers/PBiCGStab/PBiCGStab.C)
#include REF_FILE(solvers/smoothSolver/smoothSolver.C)

// C entry point: kind 0 PCG, 1 PBiCG, 2 PBiCGStab, 3 smoothSolver.  eng/mat: engine handles (coefficients bound); stream: the
// context's stream; psi/source: device pointers, caller order.  out5 = {initialResidual, finalResidual, nIterations, converged, singular}
extern "C" void ref_dropin_solve(int kind, mi_ctx_t eng, mi_matrix_t mat, void* stream, int n_cells, double* psi_dev, const double* source_dev,
                                 int precond, double tolerance, double relTol, int maxIter, int minIter, int n_sweeps, double omega, double* out5)
{
    using namespace Foam;
    ctx.eng = eng; ctx.mat = mat; ctx.precond = precond; ctx.omega = omega; ctx.stream = (hipStream_t)stream;
    lduMatrix A(n_cells); FieldField<gpuField, scalar> b, i; lduInterfaceFieldPtrsList ifs; dictionary d;
    scalargpuField x(psi_dev, n_cells), s(const_cast<double*>(source_dev), n_cells);
    solverPerformance sp;
#define RUN(S) S.maxIter_ = maxIter; S.minIter_ = minIter; S.tolerance_ = tolerance; S.relTol_ = relTol; sp = S.solve(x, s)
    if (kind == 0) { PCG S("p", A, b, i, ifs, d); RUN(S); }
    else if (kind == 1) { PBiCG S("U", A, b, i, ifs, d); RUN(S); }
    else if (kind == 3) { smoothSolver S("p", A, b, i, ifs, d); S.nSweeps_ = n_sweeps; RUN(S); }
    else { PBiCGStab S("U", A, b, i, ifs, d); RUN(S); }
#undef RUN
    SHIM_HIP(hipStreamSynchronize(ctx.stream));
    out5[0] = sp.initialResidual(); out5[1] = sp.finalResidual(); out5[2] = sp.nIterations(); out5[3] = sp.converged(); out5[4] = sp.singular();
}
