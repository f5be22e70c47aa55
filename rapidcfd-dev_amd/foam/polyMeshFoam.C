// polyMeshFoam.C -- laplacianFoam's pressure-like step on a real OpenFOAM case directory: read constant/polyMesh
// (ascii or binary), build the geometric face fields, assemble fvm::laplacian(p) == S with fixedValue 0 on the patches
// of type `patch` and zeroGradient on the `wall` ones, solve through lduMatrix::solver::New.  The mesh/IO layer is the
// caller of the hot path; this application shows the path fed from OpenFOAM's own on-disk format instead of the synthetic
// box (tests/test_polymesh.py writes the case, recomputes geometry and coefficients with numpy and checks every line).
//
// usage: polyMeshFoam <caseDir>      (source term: <caseDir>/0/S, volScalarField)
#include "polyMesh.H"

#include <cmath>
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
        {
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "PCG"}, {"preconditioner", "DIC"}, {"tolerance", "1e-09"}, {"relTol", "0"}});
        }
        {
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "GAMG"}, {"smoother", "GaussSeidel"}, {"agglomerator", "faceAreaPair"}, {"nCellsInCoarsestLevel", "10"},
                                       {"mergeLevels", "1"}, {"tolerance", "1e-09"}, {"relTol", "0"}, {"cacheAgglomeration", "true"}});
            std::vector<scalar> h = psi.asHost();
            scalar s = 0, m = 0;
            for (scalar v : h) { s += v; m = std::max(m, std::fabs(v)); }
            Info << "p sum max: " << s << " " << m << std::endl;
        }
        Info << "End" << std::endl;
        return 0;
    } catch (const Foam::error& e) { std::cerr << e.what() << std::endl; return 1; }
}
