// Translation unit that compiles the REFERENCE's PCG.C, PBiCG.C and PBiCGStab.C (and its two functor headers) where they lie.
// The shim owns the include guards of the class headers those files ask for.  REF_LDU = ".../src/OpenFOAM/matrices/lduMatrix"
#include "foam_solver_shim.H"
#define PCG_H
#define PBiCG_H
#define PBiCGStab_H
#define PCGCache_H
#define smoothSolver_H
#define lduMatrix_H
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
namespace Foam { refContext ctx = {0, 1, 0.9}; int lduMatrix::debug = 0;
const scalar solverPerformance::great_ = 1e20; const scalar solverPerformance::small_ = 1e-20; const scalar solverPerformance::vsmall_ = 1e-300; }
#include REF_FILE(lduMatrix/lduMatrixSolverFunctors.H)
#include REF_FILE(lduMatrix/lduMatrixFunctors.H)
#include REF_FILE(solvers/PCG/PCG.C)
#include REF_FILE(solvers/PBiCG/PBiCG.C)
#include REF_FILE(solvers/PBiCGStab/PBiCGStab.C)
#include REF_FILE(solvers/smoothSolver/smoothSolver.C)

// C entry point: kind 0 PCG, 1 PBiCG, 2 PBiCGStab, 3 smoothSolver (n_sweeps, omega); out5 = {initialResidual, finalResidual, nIterations, converged, singular}
extern "C" void ref_krylov_solve(int kind, const orc_system* sys, double* psi, const double* source, int precond, double tolerance,
                                 double relTol, int maxIter, int minIter, int n_sweeps, double omega, double* out5)
{
    using namespace Foam;
    ctx.sys = sys; ctx.precond = precond; ctx.omega = omega;
    const label n = (label)sys->nTotal;
    lduMatrix A; FieldField<gpuField, scalar> b, i; lduInterfaceFieldPtrsList ifs; dictionary d;
    scalargpuField x(psi, n), s(const_cast<double*>(source), n);
    solverPerformance* sp = 0;
    if (kind == 0) { PCG S("p", A, b, i, ifs, d); S.maxIter_ = maxIter; S.minIter_ = minIter; S.tolerance_ = tolerance; S.relTol_ = relTol; sp = new solverPerformance(S.solve(x, s)); }
    else if (kind == 1) { PBiCG S("U", A, b, i, ifs, d); S.maxIter_ = maxIter; S.minIter_ = minIter; S.tolerance_ = tolerance; S.relTol_ = relTol; sp = new solverPerformance(S.solve(x, s)); }
    else if (kind == 3) { smoothSolver S("p", A, b, i, ifs, d); S.nSweeps_ = n_sweeps; S.maxIter_ = maxIter; S.minIter_ = minIter; S.tolerance_ = tolerance; S.relTol_ = relTol; sp = new solverPerformance(S.solve(x, s)); }
    else { PBiCGStab S("U", A, b, i, ifs, d); S.maxIter_ = maxIter; S.minIter_ = minIter; S.tolerance_ = tolerance; S.relTol_ = relTol; sp = new solverPerformance(S.solve(x, s)); }
    out5[0] = sp->initialResidual(); out5[1] = sp->finalResidual(); out5[2] = sp->nIterations(); out5[3] = sp->converged(); out5[4] = sp->singular();
    delete sp;
}
