// Translation unit that compiles the REFERENCE's solvers/GAMG/GAMGSolverSolve.C where it lies, for the GPU, on top of the
// engine's level operators (see foam_engine_gamg_shim.H).
#include "foam_engine_gamg_shim.H"
#define GAMGSolver_H
#define REF_STR2(x) #x
#define REF_STR(x) REF_STR2(x)
#define REF_FILE(rel) REF_STR(REF_LDU/rel)
namespace Foam { label UPstream::warnComm = -1; const word GAMGSolver::typeName("GAMG"); int GAMGSolver::debug = 0; }
#include REF_FILE(solvers/GAMG/GAMGSolverSolve.C)

struct gamg_controls_c { double tolerance, relTol; int32_t maxIter, minIter, nPreSweeps, preSweepsLevelMultiplier, maxPreSweeps, nPostSweeps, postSweepsLevelMultiplier, maxPostSweeps, nFinestSweeps, scaleCorrection; double omega; int32_t directSolveCoarsest, reserved; };

// C entry point: the reference's GAMGSolver::solve with the engine's hierarchy g over matrix mat; out5 as in ref_dropin_solve
extern "C" void ref_dropin_gamg_solve(mi_ctx_t eng, mi_gamg_t g, mi_matrix_t mat, void* stream, int n_cells, int asym, double* psi_dev,
                                      const double* source_dev, const gamg_controls_c* c, double* out5)
{
    using namespace Foam;
    ctx.eng = eng; ctx.mat = mat; ctx.omega = c->omega; ctx.stream = (hipStream_t)stream; ctx.asym = asym != 0;
    SHIM_MI(mi_gamg_update(g, mat));                                                                      // GAMGSolver.C:88-172
    const int nL = mi_gamg_n_levels(g);
    GAMGAgglomeration agg; agg.g = g;
    lduMatrix A(n_cells); FieldField<gpuField, scalar> b, i; lduInterfaceFieldPtrsList ifs; dictionary d;
    GAMGSolver G("p", A, b, i, ifs, d, agg);
    G.maxIter_ = c->maxIter; G.minIter_ = c->minIter; G.tolerance_ = c->tolerance; G.relTol_ = c->relTol;
    G.nPreSweeps_ = c->nPreSweeps; G.preSweepsLevelMultiplier_ = c->preSweepsLevelMultiplier; G.maxPreSweeps_ = c->maxPreSweeps;
    G.nPostSweeps_ = c->nPostSweeps; G.postSweepsLevelMultiplier_ = c->postSweepsLevelMultiplier; G.maxPostSweeps_ = c->maxPostSweeps;
    G.nFinestSweeps_ = c->nFinestSweeps; G.interpolateCorrection_ = false; G.directSolveCoarsest_ = true; G.cacheAgglomeration_ = true;
    G.scaleCorrection_ = c->scaleCorrection < 0 ? !ctx.asym : (c->scaleCorrection != 0);                  // GAMGSolver.C:76
    G.matrixLevels_.setSize(nL); G.interfaceLevels_.setSize(nL); G.interfaceLevelsBouCoeffs_.setSize(nL); G.interfaceLevelsIntCoeffs_.setSize(nL);
    for (int l = 0; l < nL; l++) {
        mi_matrix_t lm = 0; SHIM_MI(mi_gamg_level_matrix(g, l, &lm));
        G.matrixLevels_.set(l, new lduMatrix(agg.nCells(l), lm));
        G.interfaceLevels_.set(l, new lduInterfaceFieldPtrsList);
        G.interfaceLevelsBouCoeffs_.set(l, new FieldField<gpuField, scalar>); G.interfaceLevelsIntCoeffs_.set(l, new FieldField<gpuField, scalar>);
    }
    LUscalarMatrix* LU = new LUscalarMatrix; LU->g = g; G.coarsestLUMatrixPtr_.set(LU);
    scalarField buffer((std::size_t)agg.nCells(nL - 1)); G.coarsestBufferPtr_ = &buffer;
    scalargpuField x(psi_dev, n_cells), s(const_cast<double*>(source_dev), n_cells);
    solverPerformance sp = G.solve(x, s);
    SHIM_HIP(hipStreamSynchronize(ctx.stream));
    out5[0] = sp.initialResidual(); out5[1] = sp.finalResidual(); out5[2] = sp.nIterations(); out5[3] = sp.converged(); out5[4] = sp.singular();
}
