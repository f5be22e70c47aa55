// Translation unit that compiles the REFERENCE's solvers/GAMG/GAMGSolverSolve.C where it lies (see foam_gamg_shim.H).
#include "foam_gamg_shim.H"
#define GAMGSolver_H
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
namespace Foam { refContext ctx = {0, 1, 0.9}; int lduMatrix::debug = 0; label UPstream::warnComm = -1;
const scalar solverPerformance::great_ = 1e20; const scalar solverPerformance::small_ = 1e-20; const scalar solverPerformance::vsmall_ = 1e-300;
const word GAMGSolver::typeName("GAMG"); int GAMGSolver::debug = 0; }
#include REF_FILE(solvers/GAMG/GAMGSolverSolve.C)

extern "C" {
orc_system *orc_gamg_sys_coarse_system(const gamg_sys_hier *H, int l, const orc_system *F);
orc_lu *orc_gamg_sys_coarsest_lu(const orc_system *Ac);
void orc_lu_free(orc_lu *L);
void orc_sys_destroy(orc_system *s);
int orc_gamg_sys_n_levels(const gamg_sys_hier *H);
}
struct gamg_controls_c { double tolerance, relTol; int32_t maxIter, minIter, nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps, nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps, nFinestSweeps, scaleCorrection; double omega; int32_t directSolveCoarsest, reserved; };

// C entry point: the reference's GAMGSolver::solve on the oracle's hierarchy H and fine system S; out5 as in ref_krylov_solve
extern "C" void ref_gamg_solve(const gamg_sys_hier* H, const orc_system* S, double* psi, const double* source, const gamg_controls_c* c, double* out5)
{
    using namespace Foam;
    const int nL = orc_gamg_sys_n_levels(H);
    ctx.sys = S; ctx.omega = c->omega;
    GAMGAgglomeration agg; agg.H = H; agg.sys.push_back(S);
    for (int l = 0; l < nL; l++) agg.sys.push_back(orc_gamg_sys_coarse_system(H, l, agg.sys.back()));   // GAMGSolver.C:88-97
    lduMatrix A(S); FieldField<gpuField, scalar> b, i; lduInterfaceFieldPtrsList ifs; dictionary d;
    GAMGSolver G("p", A, b, i, ifs, d, agg);
    G.maxIter_ = c->maxIter; G.minIter_ = c->minIter; G.tolerance_ = c->tolerance; G.relTol_ = c->relTol;
    G.nPreSweeps_ = c->nPreSweeps; G.preSweepsLevelMultiplier_ = c->preSweepsLevelMultiplier; G.maxPreSweeps_ = c->maxPreSweeps;
    G.nPostSweeps_ = c->nPostSweeps; G.postSweepsLevelMultiplier_ = c->postSweepsLevelMultiplier; G.maxPostSweeps_ = c->maxPostSweeps;
    G.nFinestSweeps_ = c->nFinestSweeps; G.interpolateCorrection_ = false; G.directSolveCoarsest_ = true; G.cacheAgglomeration_ = true;
    G.scaleCorrection_ = c->scaleCorrection < 0 ? !A.asymmetric() : (c->scaleCorrection != 0);                 // GAMGSolver.C:76
    G.matrixLevels_.setSize(nL); G.interfaceLevels_.setSize(nL); G.interfaceLevelsBouCoeffs_.setSize(nL); G.interfaceLevelsIntCoeffs_.setSize(nL);
    for (int l = 0; l < nL; l++) {
        G.matrixLevels_.set(l, new lduMatrix(agg.sys[(std::size_t)l + 1]));
        G.interfaceLevels_.set(l, new lduInterfaceFieldPtrsList);
        G.interfaceLevelsBouCoeffs_.set(l, new FieldField<gpuField, scalar>); G.interfaceLevelsIntCoeffs_.set(l, new FieldField<gpuField, scalar>);
    }
    orc_lu* lu = orc_gamg_sys_coarsest_lu(agg.sys.back());                                                      // GAMGSolver.C:144-172
    LUscalarMatrix* LU = new LUscalarMatrix; LU->lu = lu; G.coarsestLUMatrixPtr_.set(LU);
    scalarField buffer((std::size_t)agg.sys.back()->nTotal); G.coarsestBufferPtr_ = &buffer;
    const label n = (label)S->nTotal;
    scalargpuField x(psi, n), s(const_cast<double*>(source), n);
    solverPerformance sp = G.solve(x, s);
    out5[0] = sp.initialResidual(); out5[1] = sp.finalResidual(); out5[2] = sp.nIterations(); out5[3] = sp.converged(); out5[4] = sp.singular();
    orc_lu_free(lu);
    for (int l = 1; l <= nL; l++) orc_sys_destroy(const_cast<orc_system*>(agg.sys[(std::size_t)l]));
}
