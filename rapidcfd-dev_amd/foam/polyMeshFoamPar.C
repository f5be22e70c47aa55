// polyMeshFoamPar.C -- polyMeshFoam on a DECOMPOSED case: every rank (one per GPU; RANK / WORLD_SIZE in the environment,
// MI_COMM_ID_FILE = a path all ranks see) reads its own <caseDir>/processor<rank>/constant/polyMesh as decomposePar wrote
// it, turns the `processor` patches into processorLduInterfaceFields, receives the neighbours' cell centres through
// coupledFvPatchField::patchNeighbourField (the engine's halo exchange) to build the coupled deltaCoeffs
// (processorFvPatch.C:42-110), assembles fvm::laplacian(p) == S (fixedValue 0 on `patch` patches, zeroGradient on `wall`
// ones, coupled coefficients -|Sf|*deltaCoeffs on the processor patches: gaussLaplacianScheme.C:60-88 with
// coupledFvPatchField::gradientInternalCoeffs/BoundaryCoeffs) and solves the global system through
// lduMatrix::solver::New(fieldName, matrix, bouCoeffs, intCoeffs, interfaces, dict), fvScalarMatrix.C:170-178.
//
//   polyMeshFoamPar <caseDir>              the solve (all ranks together)
//   polyMeshFoamPar <caseDir> -check <p>   read processor<p> only and print its mesh statistics (no device, no communicator)
//
// Processor patches pair up by (myProcNo, neighbProcNo): the k-th patch towards rank q matches the k-th patch of q towards
// this rank.  A patch whose neighbour is the rank ITSELF (not something decomposePar produces; tests use it to send a
// periodic coupling through RCCL on one GPU) pairs with the next / previous self-patch.
#include "polyMesh.H"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iomanip>

using namespace Foam;

int main(int argc, char** argv)
{
    try {
        if (argc < 2) { std::cerr << "usage: polyMeshFoamPar <caseDir> [-check procNo]" << std::endl; return 2; }
        const std::string caseDir = argv[1];
        const bool check = argc >= 4 && std::strcmp(argv[2], "-check") == 0;
        const int rank = check ? atoi(argv[3]) : (getenv("RANK") ? atoi(getenv("RANK")) : 0);
        const int world = check ? 1 : (getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1);
        polyMesh mesh(caseDir, rank);
        const label n = mesh.nCells, nI = mesh.nInternalFaces();
        Info << std::setprecision(17);
        if (check) {
            scalar sumV = 0; for (scalar v : mesh.V) sumV += v;
            Info << "processor" << rank << ": nPoints " << mesh.points.size() << " nCells " << n << " nFaces " << mesh.nFaces() << " nInternalFaces " << nI << " sumV " << sumV << std::endl;
            for (std::size_t p = 0; p < mesh.boundary.size(); ++p) {
                const polyPatch& P = mesh.boundary[p];
                scalar a = 0; for (scalar v : mesh.patchMagSf[p]) a += v;
                Info << "patch " << P.name << " type " << P.type << " nFaces " << P.nFaces << " startFace " << P.startFace << " myProcNo " << P.myProcNo
                     << " neighbProcNo " << P.neighbProcNo << " area " << a << std::endl;
            }
            Info << "End" << std::endl;
            return 0;
        }
        const char* idf = getenv("MI_COMM_ID_FILE");
        Pstream::init(rank, world, idf ? idf : "/tmp/mi_comm_id");
        const bool talk = Pstream::master();

        // ---- coupled patches of this processor ----
        std::vector<label> procPatches;
        for (label p = 0; p < (label)mesh.boundary.size(); ++p) if (mesh.boundary[(std::size_t)p].type == "processor") procPatches.push_back(p);
        std::vector<lduInterface> ifs;
        for (std::size_t k = 0; k < procPatches.size(); ++k) {
            const polyPatch& P = mesh.boundary[(std::size_t)procPatches[k]];
            if (P.myProcNo != rank) FatalErrorIn("polyMeshFoamPar", "patch " + P.name + " belongs to another processor");
            lduInterface I; I.type = "processor"; I.faceCells = mesh.patchFaceCells(procPatches[k]); I.neighbProcNo = P.neighbProcNo;
            label ord = 0;                                   // ordinal among my patches towards the same neighbour
            for (std::size_t q = 0; q < k; ++q) if (mesh.boundary[(std::size_t)procPatches[q]].neighbProcNo == P.neighbProcNo) ++ord;
            I.neighbPatchID = P.neighbProcNo == rank ? (ord ^ 1) : ord;
            ifs.push_back(I);
        }
        labelList lower(mesh.owner.begin(), mesh.owner.begin() + nI);
        lduAddressing addr(n, lower, mesh.neighbour, ifs);
        lduInterfaceFieldPtrsList interfaces;
        for (const lduInterface& I : ifs) interfaces.push_back(new processorLduInterfaceField(I.faceCells, rank, I.neighbProcNo, I.neighbPatchID));

        // ---- neighbour cell centres through the halo exchange -> coupled deltaCoeffs ----
        lduMatrix A(addr);
        FieldFieldScalar bouCoeffs, intCoeffs;
        for (const lduInterface& I : ifs) { bouCoeffs.push_back(scalargpuField((label)I.faceCells.size())); intCoeffs.push_back(scalargpuField((label)I.faceCells.size())); }
        A.diag() = scalarField((std::size_t)n, 1.0); A.upper() = scalarField((std::size_t)nI, 0.0);
        std::vector<vectorField> Cn(ifs.size());
        for (std::size_t k = 0; k < ifs.size(); ++k) Cn[k].assign(ifs[k].faceCells.size(), vector{0, 0, 0});
        for (int d = 0; d < 3; ++d) {
            scalarField comp((std::size_t)n);
            for (label c = 0; c < n; ++c) comp[(std::size_t)c] = mesh.C[(std::size_t)c][(std::size_t)d];
            FieldFieldScalar nbr;
            A.patchNeighbourField(nbr, scalargpuField(comp), bouCoeffs, interfaces);
            for (std::size_t k = 0; k < ifs.size(); ++k) { const std::vector<scalar> h = nbr[k].asHost(); for (std::size_t i = 0; i < h.size(); ++i) Cn[k][i][(std::size_t)d] = h[i]; }
        }
        scalar sumDc = 0, sumW = 0;
        std::vector<scalarField> bou(ifs.size());
        for (std::size_t k = 0; k < ifs.size(); ++k) {
            scalarField dc, w;
            mesh.coupledPatchGeometry(procPatches[k], Cn[k], dc, w);
            bou[k].resize(dc.size());
            for (std::size_t i = 0; i < dc.size(); ++i) { bou[k][i] = -(mesh.patchMagSf[(std::size_t)procPatches[k]][i] * dc[i]); sumDc += dc[i]; sumW += w[i]; }
        }
        const scalar gDc = Pstream::returnReduceSum(sumDc), gW = Pstream::returnReduceSum(sumW);
        scalar sumV = 0; for (scalar v : mesh.V) sumV += v;
        const scalar gV = Pstream::returnReduceSum(sumV);
        if (talk) {
            Info << "Create mesh for " << world << " processor(s): global sumV " << gV << std::endl;
            Info << "coupled geometry: sumDeltaCoeffs " << gDc << " sumWeights " << gW << std::endl;
        }

        // ---- fvm::laplacian(p) == S ----
        scalarField upper((std::size_t)nI), diag((std::size_t)n, 0.0);
        for (label f = 0; f < nI; ++f) {
            upper[(std::size_t)f] = mesh.nonOrthDeltaCoeffs[(std::size_t)f] * mesh.magSf[(std::size_t)f];
            diag[(std::size_t)mesh.owner[(std::size_t)f]] -= upper[(std::size_t)f]; diag[(std::size_t)mesh.neighbour[(std::size_t)f]] -= upper[(std::size_t)f];
        }
        for (label p = 0; p < (label)mesh.boundary.size(); ++p) {       // addBoundaryDiag: fixedValue patches
            if (mesh.boundary[(std::size_t)p].type != "patch") continue;
            const labelList fc = mesh.patchFaceCells(p);
            for (std::size_t i = 0; i < fc.size(); ++i) diag[(std::size_t)fc[i]] += -(mesh.patchMagSf[(std::size_t)p][i] * mesh.patchDeltaCoeffs[(std::size_t)p][i]);
        }
        for (std::size_t k = 0; k < ifs.size(); ++k)                        // ... and the coupled patches' internalCoeffs
            for (std::size_t i = 0; i < bou[k].size(); ++i) diag[(std::size_t)ifs[k].faceCells[i]] += bou[k][i];
        A.diag() = diag; A.upper() = upper;
        bouCoeffs.clear(); intCoeffs.clear();
        for (std::size_t k = 0; k < ifs.size(); ++k) { bouCoeffs.push_back(scalargpuField(bou[k])); intCoeffs.push_back(scalargpuField(bou[k])); }
        scalarField S = readVolScalarInternalField(caseDir + "/processor" + std::to_string(rank) + "/0/S", n);
        for (label c = 0; c < n; ++c) S[(std::size_t)c] *= mesh.V[(std::size_t)c];
        scalargpuField source(S);
        const scalarField w = mesh.faceAreaPairWeights();
        setFaceAreaPairWeights(&w);
        const dictionary dicts[] = {
            dictionary{{"solver", "PCG"}, {"preconditioner", "DIC"}, {"tolerance", "1e-09"}, {"relTol", "0"}},
            dictionary{{"solver", "GAMG"}, {"smoother", "GaussSeidel"}, {"agglomerator", "faceAreaPair"}, {"nCellsInCoarsestLevel", "10"},
                       {"mergeLevels", "1"}, {"tolerance", "1e-09"}, {"relTol", "0"}, {"cacheAgglomeration", "true"}},
        };
        scalar s = 0, m = 0;
        for (const dictionary& d : dicts) {
            scalargpuField psi(n);
            solverPerformance sp = lduMatrix::solver::New("p", A, bouCoeffs, intCoeffs, interfaces, d)->solve(psi, source);
            if (talk) sp.print(Info);
            s = 0; m = 0;
            for (scalar v : psi.asHost()) { s += v; m = std::max(m, std::fabs(v)); }
        }
        const scalar gs = Pstream::returnReduceSum(s);
        if (talk) Info << "p sum (global) max (rank 0): " << gs << " " << m << std::endl;
        for (auto* f : interfaces) delete f;
        Pstream::exit();
        if (talk) Info << "End" << std::endl;
        return 0;
    } catch (const Foam::error& e) { std::cerr << e.what() << std::endl; return 1; }
}
