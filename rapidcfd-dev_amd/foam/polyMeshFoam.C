// polyMeshFoam.C -- laplacianFoam's pressure-like step on a real OpenFOAM case directory: read constant/polyMesh
// (ascii or binary), build the geometric face fields, assemble fvm::laplacian(p) == S with fixedValue 0 on the patches
// of type `patch` and zeroGradient on the `wall` ones, solve through lduMatrix::solver::New.  The mesh/IO layer is the
// caller of the hot path; this application shows the path fed from OpenFOAM's own on-disk format instead of the synthetic
// box (tests/test_polymesh.py writes the case, recomputes geometry and coefficients with numpy and checks every line).
//
// usage: polyMeshFoam <caseDir> [-nonOrthCorrectors N] [-write <timeName> [-writeFormat ascii|binary] [-writePrecision N]]
//        (source term: <caseDir>/0/S, volScalarField; -write: the GAMG solution goes back into the case as <caseDir>/<timeName>/p)
//        With a <caseDir>/system/fvSolution the pressure solves take their controls from it, as fvMatrix::solve() does:
//        solvers.p for the first solve, solvers.pFinal (if present, else solvers.p) for the second -- without one: PCG + DIC, then GAMG.
//        polyMeshFoam <caseDir> -solverDict <fieldName>      (host only: the controls lduMatrix::solver::New would be handed, and the
//                                                             relaxation factors of that name, read from <caseDir>/system/fvSolution)
//        polyMeshFoam <caseDir> -scheme <ddt|div|grad|laplacian|interpolation|snGrad|d2dt2> <name>   (host only: <caseDir>/system/fvSchemes)
//        polyMeshFoam <caseDir> -roundTrip <object> <timeName> <ascii|binary> <precision>
//        (host only: <caseDir>/0/<object> is read, written as <caseDir>/<timeName>/<object>, read again and compared)
// -nonOrthCorrectors N: afterwards, laplacianFoam's non-orthogonal corrector loop (laplacianFoam.C:60-70) with the `corrected`
// snGrad scheme: N times { assemble fvm::laplacian incl. the explicit correction from the current p; solve with PCG + DIC }.
#include "polyMesh.H"
#include "solution.H"

#include <cmath>
#include <fstream>
#include <iomanip>

using namespace Foam;

int main(int argc, char** argv)
{
    try {
        if (argc < 2) { std::cerr << "usage: polyMeshFoam <caseDir>" << std::endl; return 2; }
        const std::string caseDir = argv[1];
        polyMesh mesh(caseDir);
        const label n = mesh.nCells, nI = mesh.nInternalFaces();
        Info << std::setprecision(17);
        for (int k = 2; k + 1 < argc; ++k) if (std::string(argv[k]) == "-solverDict") {   // dictionary I/O only: no device
            const solution sol(caseDir);
            const word field = argv[k + 1];
            const dictionary d = sol.solverDict(field);
            Info << "solvers." << field << std::endl;
            for (const auto& kv : d.entries()) Info << "    " << kv.first << " = " << kv.second << std::endl;
            Info << "relaxField " << sol.relaxField(field) << " relaxEquation " << sol.relaxEquation(field) << std::endl;
            if (sol.relaxField(field)) Info << "fieldRelaxationFactor " << sol.fieldRelaxationFactor(field) << std::endl;
            if (sol.relaxEquation(field)) Info << "equationRelaxationFactor " << sol.equationRelaxationFactor(field) << std::endl;
            for (const char* alg : {"PISO", "SIMPLE", "PIMPLE"}) if (sol.solutionDict().found(alg)) {
                Info << alg << std::endl;
                const dictionary ad = sol.dict(alg);
                for (const auto& kv : ad.entries()) Info << "    " << kv.first << " = " << kv.second << std::endl;
            }
            Info << "End" << std::endl;
            return 0;
        }
        for (int k = 2; k + 2 < argc; ++k) if (std::string(argv[k]) == "-scheme") {      // <kind> <name>: what mesh.<kind>Scheme(name) returns
            const fvSchemes sch(caseDir);
            const word kind = argv[k + 1], name = argv[k + 2];
            wordList t;
            if (kind == "ddt") t = sch.ddtScheme(name); else if (kind == "div") t = sch.divScheme(name); else if (kind == "grad") t = sch.gradScheme(name);
            else if (kind == "laplacian") t = sch.laplacianScheme(name); else if (kind == "interpolation") t = sch.interpolationScheme(name);
            else if (kind == "snGrad") t = sch.snGradScheme(name); else if (kind == "d2dt2") t = sch.d2dt2Scheme(name);
            else FatalErrorIn("polyMeshFoam -scheme", "unknown kind " + kind);
            Info << kind << "Scheme(" << name << ") =";
            for (const word& w : t) Info << " " << w;
            Info << std::endl << "steady " << sch.steady() << " fluxRequired " << sch.fluxRequired(name) << std::endl << "End" << std::endl;
            return 0;
        }
        for (int k = 2; k + 4 < argc; ++k) if (std::string(argv[k]) == "-roundTrip") {   // field I/O only: no device, no engine context
            const std::string obj = argv[k + 1], time = argv[k + 2];
            const bool bin = std::string(argv[k + 3]) == "binary";
            const int prec = std::atoi(argv[k + 4]);
            const scalarField in = readVolScalarInternalField(caseDir + "/0/" + obj, n);
            std::vector<patchFieldOut> bf;
            for (label p = 0; p < (label)mesh.boundary.size(); ++p) {
                patchFieldOut e; e.patchName = mesh.boundary[p].name;
                if (mesh.boundary[p].type == "patch") { e.type = "fixedValue"; e.hasValue = true; e.value = scalarField((std::size_t)mesh.boundary[p].nFaces, 0.25 * (p + 1)); }
                else if (mesh.boundary[p].type == "processor") { e.type = "processor"; e.hasValue = true; e.value.resize((std::size_t)mesh.boundary[p].nFaces); for (std::size_t i = 0; i < e.value.size(); ++i) e.value[i] = in[(std::size_t)mesh.patchFaceCells(p)[i]]; }
                else e.type = "zeroGradient";
                bf.push_back(e);
            }
            writeVolScalarField(caseDir, time, obj, "[0 0 -1 0 0 0 0]", in, bf, bin, prec);
            const scalarField back = readVolScalarInternalField(caseDir + "/" + time + "/" + obj, n);
            scalar d = 0;
            for (label c = 0; c < n; ++c) d = std::max(d, std::fabs(back[c] - in[c]));
            vectorField U((std::size_t)n);                                              // ... and a vector field: the cell centres
            for (label c = 0; c < n; ++c) U[c] = mesh.C[c];
            std::vector<patchFieldOut> bfU;
            for (const polyPatch& P : mesh.boundary) { patchFieldOut e; e.patchName = P.name; e.type = "zeroGradient"; bfU.push_back(e); }
            writeVolVectorField(caseDir, time, "C", "[0 1 0 0 0 0 0]", U, bfU, bin, prec);
            const vectorField Ub = readVolVectorInternalField(caseDir + "/" + time + "/C", n);
            scalar dU = 0;
            for (label c = 0; c < n; ++c) for (int q = 0; q < 3; ++q) dU = std::max(dU, std::fabs(Ub[c][q] - U[c][q]));
            Info << "roundTrip " << obj << " nCells " << n << " maxAbsDiff " << d << " vector maxAbsDiff " << dU << std::endl << "End" << std::endl;
            return 0;
        }
        Info << "Create mesh: nPoints " << mesh.points.size() << " nCells " << n << " nFaces " << mesh.nFaces() << " nInternalFaces " << nI << std::endl;
        scalar sumV = 0, sumMagSf = 0, sumW = 0, sumD = 0;
        for (scalar v : mesh.V) sumV += v;
        for (label f = 0; f < nI; ++f) { sumMagSf += mesh.magSf[f]; sumW += mesh.weights[f]; sumD += mesh.nonOrthDeltaCoeffs[f]; }
        Info << "geometry: sumV " << sumV << " sumMagSfInternal " << sumMagSf << " sumWeights " << sumW << " sumNonOrthDeltaCoeffs " << sumD << std::endl;
        for (const polyPatch& P : mesh.boundary) Info << "patch " << P.name << " type " << P.type << " nFaces " << P.nFaces << " startFace " << P.startFace << std::endl;

        labelList lower(mesh.owner.begin(), mesh.owner.begin() + nI);
        lduAddressing addr(n, lower, mesh.neighbour);
        std::vector<labelList> patches;
        for (label p = 0; p < (label)mesh.boundary.size(); ++p) patches.push_back(mesh.patchFaceCells(p));
        fvScalarMatrix pEqn("p", addr, patches, std::vector<bool>(patches.size(), false));
        scalarField magSfI(mesh.magSf.begin(), mesh.magSf.begin() + nI);
        fvm::laplacian(pEqn, scalargpuField(mesh.nonOrthDeltaCoeffs), scalargpuField(magSfI));   // gamma = 1
        for (std::size_t p = 0; p < patches.size(); ++p) {
            if (mesh.boundary[p].type != "patch") continue;      // wall: zeroGradient, coefficients stay zero
            scalarField ic(patches[p].size());
            for (std::size_t i = 0; i < ic.size(); ++i) ic[i] = -(mesh.patchMagSf[p][i] * mesh.patchDeltaCoeffs[p][i]); // fixedValue: gamma*|Sf|*(-deltaCoeffs)
            pEqn.internalCoeffs()[p] = ic;
        }
        scalarField S = readVolScalarInternalField(caseDir + "/0/S", n);
        for (label c = 0; c < n; ++c) S[c] *= mesh.V[c];
        pEqn.source() = S;
        const scalarField w = mesh.faceAreaPairWeights();
        setFaceAreaPairWeights(&w);
        // the solver controls: solvers.p / solvers.pFinal of <caseDir>/system/fvSolution when the case has one (fvMatrixSolve.C:56-101)
        dictionary first{{"solver", "PCG"}, {"preconditioner", "DIC"}, {"tolerance", "1e-09"}, {"relTol", "0"}};
        dictionary second{{"solver", "GAMG"}, {"smoother", "GaussSeidel"}, {"agglomerator", "faceAreaPair"}, {"nCellsInCoarsestLevel", "10"},
                          {"mergeLevels", "1"}, {"tolerance", "1e-09"}, {"relTol", "0"}, {"cacheAgglomeration", "true"}};
        {
            std::ifstream probe(caseDir + "/system/fvSolution");
            if (probe) {
                const solution sol(caseDir);
                first = sol.solverDict("p");
                second = sol.solutionDict().subDict("solvers").found("pFinal") ? sol.solverDict("pFinal") : first;
                Info << "solver controls from " << caseDir << "/system/fvSolution" << std::endl;
            }
        }
        {
            scalargpuField psi(n);
            pEqn.solve(psi, first);
        }
        {
            scalargpuField psi(n);
            pEqn.solve(psi, second);
            std::vector<scalar> h = psi.asHost();
            scalar s = 0, m = 0;
            for (scalar v : h) { s += v; m = std::max(m, std::fabs(v)); }
            Info << "p sum max: " << s << " " << m << std::endl;
            // -write <timeName>: the solution as a volScalarField of the case (fixedValue 0 on `patch`, zeroGradient on `wall`)
            std::string timeName, fmt = "ascii"; int prec = 6;
            for (int k = 2; k + 1 < argc; ++k) {
                if (std::string(argv[k]) == "-write") timeName = argv[k + 1];
                if (std::string(argv[k]) == "-writeFormat") fmt = argv[k + 1];
                if (std::string(argv[k]) == "-writePrecision") prec = std::atoi(argv[k + 1]);
            }
            if (!timeName.empty()) {
                std::vector<patchFieldOut> bf;
                for (const polyPatch& P : mesh.boundary) {
                    patchFieldOut e; e.patchName = P.name;
                    if (P.type == "patch") { e.type = "fixedValue"; e.hasValue = true; e.value = scalarField((std::size_t)P.nFaces, 0.0); }
                    else e.type = "zeroGradient";
                    bf.push_back(e);
                }
                writeVolScalarField(caseDir, timeName, "p", "[0 2 -2 0 0 0 0]", h, bf, fmt == "binary", prec);
                Info << "wrote " << caseDir << "/" << timeName << "/p (" << fmt << ")" << std::endl;
            }
        }
        label nCorr = 0;
        for (int k = 2; k + 1 < argc; ++k) if (std::string(argv[k]) == "-nonOrthCorrectors") nCorr = (label)std::atoi(argv[k + 1]);
        if (nCorr > 0) {
            auto comp = [&](const vectorField& v, std::size_t b, std::size_t e) {
                vectorgpuField out((label)(e - b));
                for (direction d = 0; d < 3; ++d) { scalarField h(e - b); for (std::size_t i = b; i < e; ++i) h[i - b] = v[i][d]; out.component(d) = h; }
                return out;
            };
            const vectorgpuField SfI = comp(mesh.Sf, 0, (std::size_t)nI), corrVecs = comp(mesh.nonOrthCorrectionVectors, 0, (std::size_t)nI);
            std::vector<vectorgpuField> pSfStore;
            for (const polyPatch& P : mesh.boundary) pSfStore.push_back(comp(mesh.Sf, (std::size_t)P.startFace, (std::size_t)(P.startFace + P.nFaces)));
            std::vector<const vectorgpuField*> pSf;
            for (const vectorgpuField& v : pSfStore) pSf.push_back(&v);
            std::vector<scalargpuField> zeros;
            for (const polyPatch& P : mesh.boundary) zeros.emplace_back(P.nFaces);
            std::vector<const scalargpuField*> pv;                        // fixedValue 0 on `patch`, zeroGradient (patchInternalField) on `wall`
            for (std::size_t p = 0; p < patches.size(); ++p) pv.push_back(mesh.boundary[p].type == "patch" ? &zeros[p] : nullptr);
            const scalargpuField weights(mesh.weights), gammaMagSf(magSfI), V(mesh.V);
            scalargpuField p(n);
            for (label corr = 0; corr < nCorr; ++corr) {
                pEqn.source() = S;
                pEqn.nonOrthCorrection(p, pv, SfI, pSf, weights, corrVecs, gammaMagSf, V);
                pEqn.solve(p, dictionary{{"solver", "PCG"}, {"preconditioner", "DIC"}, {"tolerance", "1e-10"}, {"relTol", "0"}});
            }
            std::vector<scalar> h = p.asHost();
            scalar s = 0, m = 0;
            for (scalar v : h) { s += v; m = std::max(m, std::fabs(v)); }
            Info << "corrected p sum max: " << s << " " << m << std::endl;
        }
        Info << "End" << std::endl;
        return 0;
    } catch (const Foam::error& e) { std::cerr << e.what() << std::endl; return 1; }
}
