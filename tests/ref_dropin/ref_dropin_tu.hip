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
namespace Foam { refContext ctx = {0, 0, 1, 0.9, 0, false, false}; int lduMatrix::debug = 0;
const scalar solverPerformance::great_ = 1e20; const scalar solverPerformance::small_ = 1e-20; const scalar solverPerformance::vsmall_ = 1e-300; }
#include REF_FILE(lduMatrix/lduMatrixSolverFunctors.H)
#include REF_FILE(lduMatrix/lduMatrixFunctors.H)
#include REF_FILE(solvers/PCG/PCG.C)
#include REF_FILE(solvers/PBiCG/PBiCG.C)
#include REF_FILE(solvers/PBiCGStab/PBiCGStab.C)
#include REF_FILE(solvers/smoothSolver/smoothSolver.C)

// C entry point: kind 0 PCG, 1 PBiCG, 2 PBiCGStab, 3 smoothSolver.  eng/mat: engine handles (coefficients bound); stream: the
// context's stream; psi/source: device pointers, caller order.  out5 = {initialResidual, finalResidual, nIterations, converged, singular}
extern "C" void ref_dropin_solve_order(int kind, mi_ctx_t eng, mi_matrix_t mat, void* stream, int n_cells, double* psi_dev, const double* source_dev,
                                       int precond, double tolerance, double relTol, int maxIter, int minIter, int n_sweeps, double omega, int engine_order, double* out5)
{
    using namespace Foam;
    ctx.eng = eng; ctx.mat = mat; ctx.precond = precond; ctx.omega = omega; ctx.stream = (hipStream_t)stream; ctx.engineOrder = engine_order != 0;
    lduMatrix A(n_cells); FieldField<gpuField, scalar> b, i; lduInterfaceFieldPtrsList ifs; dictionary d;
    // engine-order mode (uncoupled matrices): psi and source are permuted once on the way in, psi once on the way out
    scalargpuField pe(engine_order ? n_cells : 0), se(engine_order ? n_cells : 0);
    if (engine_order) { SHIM_MI(mi_vec_to_engine(mi_matrix_addr(mat), psi_dev, pe.data())); SHIM_MI(mi_vec_to_engine(mi_matrix_addr(mat), source_dev, se.data())); }
    scalargpuField x(engine_order ? pe.data() : psi_dev, n_cells), s(engine_order ? se.data() : const_cast<double*>(source_dev), n_cells);
    solverPerformance sp;
#define RUN(S) S.maxIter_ = maxIter; S.minIter_ = minIter; S.tolerance_ = tolerance; S.relTol_ = relTol; sp = S.solve(x, s)
    if (kind == 0) { PCG S("p", A, b, i, ifs, d); RUN(S); }
    else if (kind == 1) { PBiCG S("U", A, b, i, ifs, d); RUN(S); }
    else if (kind == 3) { smoothSolver S("p", A, b, i, ifs, d); S.nSweeps_ = n_sweeps; RUN(S); }
    else { PBiCGStab S("U", A, b, i, ifs, d); RUN(S); }
#undef RUN
    if (engine_order) SHIM_MI(mi_vec_from_engine(mi_matrix_addr(mat), pe.data(), psi_dev));
    SHIM_HIP(hipStreamSynchronize(ctx.stream));
    ctx.engineOrder = false;
    out5[0] = sp.initialResidual(); out5[1] = sp.finalResidual(); out5[2] = sp.nIterations(); out5[3] = sp.converged(); out5[4] = sp.singular();
}
extern "C" void ref_dropin_solve(int kind, mi_ctx_t eng, mi_matrix_t mat, void* stream, int n_cells, double* psi_dev, const double* source_dev,
                                 int precond, double tolerance, double relTol, int maxIter, int minIter, int n_sweeps, double omega, double* out5)
{
    ref_dropin_solve_order(kind, eng, mat, stream, n_cells, psi_dev, source_dev, precond, tolerance, relTol, maxIter, minIter, n_sweeps, omega, 0, out5);
}
