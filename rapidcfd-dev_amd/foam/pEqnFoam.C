// pEqnFoam.C -- a miniature of the solver applications' pressure / momentum steps, written against
// the OpenFOAM-style interface of miFoam.H exactly the way icoFoam.C:75-92 is written against
// OpenFOAM: assemble with fvm::laplacian / fvm::div, solve through lduMatrix::solver::New with an
// fvSolution-style dictionary, print the solverPerformance line.  tests/test_foam_mirror.py runs it
// on the GPU box and checks every printed number against the oracle.
//
// usage: pEqnFoam nx ny nz
#include "miFoam.H"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>

using namespace Foam;

static double splitmixUniform(uint64_t seed, uint64_t i)
{   // rapidcfd-dev_amd/synthetic.py::splitmix_uniform
    uint64_t z = (i + 1) * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

int main(int argc, char** argv)
{
    try {
        const int nx = argc > 1 ? atoi(argv[1]) : 16, ny = argc > 2 ? atoi(argv[2]) : 16, nz = argc > 3 ? atoi(argv[3]) : 16;
        const label n = nx * ny * nz;
        const scalar h = 1.0 / nx;
        // blockMesh-like box: owner-sorted internal faces, six boundary patches
        labelList lower, upper; std::vector<int> dir;
        for (label c = 0; c < n; ++c) {
            const int i = c % nx, j = (c / nx) % ny, k = c / (nx * ny);
            if (i < nx - 1) { lower.push_back(c); upper.push_back(c + 1); dir.push_back(0); }
            if (j < ny - 1) { lower.push_back(c); upper.push_back(c + nx); dir.push_back(1); }
            if (k < nz - 1) { lower.push_back(c); upper.push_back(c + nx * ny); dir.push_back(2); }
        }
        const label nf = (label)lower.size();
        std::vector<labelList> patches(6);
        for (label c = 0; c < n; ++c) {
            const int i = c % nx, j = (c / nx) % ny, k = c / (nx * ny);
            if (i == 0) patches[0].push_back(c);
            if (i == nx - 1) patches[1].push_back(c);
            if (j == 0) patches[2].push_back(c);
            if (j == ny - 1) patches[3].push_back(c);
            if (k == 0) patches[4].push_back(c);
            if (k == nz - 1) patches[5].push_back(c);
        }
        lduAddressing addr(n, lower, upper);
        scalarField faceAreaWeights(nf);
        const double wdir[3] = {1.0, 1.01, 1.02};
        for (label f = 0; f < nf; ++f) faceAreaWeights[f] = h * wdir[dir[f]];
        setFaceAreaPairWeights(&faceAreaWeights);

        Info << std::setprecision(17);
        Info << "Create mesh: " << n << " cells, " << nf << " internal faces" << std::endl;

        // ---- pEqn: fvm::laplacian(1, p) == source, p fixedValue on patch 0 (x-min) ----
        scalarField delta(nf, 1.0 / h), gam(nf), src(n);
        for (label f = 0; f < nf; ++f) gam[f] = h * h * (1.0 + 0.1 * splitmixUniform(12345, f));
        for (label c = 0; c < n; ++c) src[c] = (2.0 * splitmixUniform(777, c) - 1.0) * h * h * h;
        fvScalarMatrix pEqn("p", addr, patches, std::vector<bool>(6, false));
        fvm::laplacian(pEqn, scalargpuField(delta), scalargpuField(gam));
        pEqn.source() = src;
        pEqn.internalCoeffs()[0] = scalarField(patches[0].size(), -2.0 * h); // fixedValue: -gamma*|Sf|*2/h
        const char* pre[] = {"diagonal", "DIC", "none"};
        for (const char* p : pre) {
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "PCG"}, {"preconditioner", p}, {"tolerance", "1e-08"}, {"relTol", "0"}});
        }
        {
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "GAMG"}, {"smoother", "GaussSeidel"}, {"agglomerator", "faceAreaPair"},
                                       {"nCellsInCoarsestLevel", "10"}, {"mergeLevels", "1"}, {"tolerance", "1e-08"}, {"relTol", "0"},
                                       {"cacheAgglomeration", "true"}});
        }
        {   // mergeLevels 2: every level folds two pair steps (GAMGAgglomeration::combineLevels); the cached hierarchy is rebuilt
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "GAMG"}, {"smoother", "GaussSeidel"}, {"agglomerator", "faceAreaPair"},
                                       {"nCellsInCoarsestLevel", "10"}, {"mergeLevels", "2"}, {"tolerance", "1e-08"}, {"relTol", "0"},
                                       {"cacheAgglomeration", "true"}});
        }
        {
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "smoothSolver"}, {"smoother", "GaussSeidel"}, {"nSweeps", "2"}, {"tolerance", "1e-03"}, {"maxIter", "400"}});
        }
        // ---- UEqn-like: ddt + fvm::div(phi) - fvm::laplacian(nu): asymmetric, PBiCG / PBiCGStab ----
        {
            scalarField wts(nf), phi(nf);
            for (label f = 0; f < nf; ++f) { phi[f] = dir[f] == 0 ? 0.3 * h * h : 0.0; wts[f] = 1.0; } // upwind, flow in +x
            fvScalarMatrix conv("Ux", addr, patches, std::vector<bool>(6, false));
            fvm::div(conv, scalargpuField(wts), scalargpuField(phi));
            // UEqn = fvm::ddt(U) + fvm::div(phi, U) - fvm::laplacian(nu, U), with the fvMatrix operators (fvMatrix.C:1693-1830)
            fvScalarMatrix UEqn("Ux", addr, patches, std::vector<bool>(6, false));
            fvm::ddt(UEqn, 1.0 / 1e-3, 1.0, scalargpuField(scalarField(n, h * h * h)), scalargpuField(n));
            UEqn += conv;
            UEqn -= pEqn;                      // brings -(-2h) = 2h into internalCoeffs of the fixedValue patch as well
            UEqn.source() = src;
            for (const char* s : {"PBiCG", "PBiCGStab"}) {
                scalargpuField psi(n);
                UEqn.solve(psi, dictionary{{"solver", s}, {"preconditioner", "DILU"}, {"tolerance", "1e-10"}, {"relTol", "0"}});
            }
            {   // rAU / HbyA ingredients of the PISO-SIMPLE corrector: UEqn.A(), UEqn.H()
                scalargpuField V(scalarField(n, h * h * h)), Aphi(n), Hphi(n), U0(src);
                UEqn.A(Aphi, V); UEqn.H(Hphi, U0, V);
                std::vector<scalar> a = Aphi.asHost(), hh = Hphi.asHost();
                scalar sa = 0, sh = 0, ma = 0, mh = 0;
                for (label c = 0; c < n; ++c) { sa += a[c]; sh += hh[c]; ma = std::max(ma, std::fabs(a[c])); mh = std::max(mh, std::fabs(hh[c])); }
                Info << "A(Ux) sum max: " << sa << " " << ma << "  H(Ux) sum max: " << sh << " " << mh << std::endl;
            }
            {   // fvMatrix::flux of the assembled UEqn for psi = src: internal faces + every boundary patch
                scalargpuField fl(nf), U0(src); FieldFieldScalar bfl;
                UEqn.flux(fl, bfl, U0);
                std::vector<scalar> f = fl.asHost();
                scalar sf = 0, mf = 0, sb = 0;
                for (label k = 0; k < nf; ++k) { sf += std::fabs(f[k]); mf = std::max(mf, std::fabs(f[k])); }
                for (const scalargpuField& b : bfl) for (scalar v : b.asHost()) sb += v;
                Info << "flux(Ux) sumMag max boundarySum: " << sf << " " << mf << " " << sb << std::endl;
            }
            {   // fvMatrix::H with a coupled patch: addBoundarySource(couples) adds boundaryCoeffs*patchNeighbourField (fvMatrix.C:318-346)
                std::vector<bool> cpl(6, false); cpl[1] = true;                                  // x-max treated as a coupled patch
                fvScalarMatrix CEqn("Ux", addr, patches, cpl);
                CEqn += UEqn;
                scalarField bc1(patches[1].size()), nbr1(patches[1].size());
                for (std::size_t i = 0; i < patches[1].size(); ++i) { bc1[i] = 0.25 * h * (1.0 + 0.5 * splitmixUniform(4242, (label)i)); nbr1[i] = splitmixUniform(4343, (label)i) - 0.5; }
                CEqn.boundaryCoeffs()[1] = bc1;
                scalargpuField V(scalarField(n, h * h * h)), Hphi(n), U0(src), nbrField(nbr1);
                std::vector<const scalargpuField*> pnf(6, nullptr); pnf[1] = &nbrField;
                CEqn.H(Hphi, U0, V, pnf);
                scalar sh = 0, mh = 0;
                for (scalar v : Hphi.asHost()) { sh += v; mh = std::max(mh, std::fabs(v)); }
                Info << "H(Ux) coupled sum max: " << sh << " " << mh << std::endl;
            }
            {   // the same UEqn written ONCE by the fused assembly (fvm::assemble -> mi_fvm_assemble): bit-identical to ddt; += conv; -= laplacian
                fvScalarMatrix FEqn("Ux", addr, patches, std::vector<bool>(6, false));
                scalarField ones(nf, 1.0), phiF(nf), dl(nf, 1.0 / h);
                for (label f = 0; f < nf; ++f) phiF[f] = dir[f] == 0 ? 0.3 * h * h : 0.0;
                scalargpuField V(scalarField(n, h * h * h)), zero(n), w(ones), ph(phiF), dlt(dl), gm(gam);
                fvm::assemble(FEqn, 1.0 / 1e-3, 1.0, V, zero, &ph, &w, &dlt, &gm);
                const bool same = FEqn.lower().asHost() == UEqn.lower().asHost() && FEqn.upper().asHost() == UEqn.upper().asHost() && FEqn.diag().asHost() == UEqn.diag().asHost();
                Info << "fvm::assemble equals the operator sequence: " << (same ? 1 : 0) << std::endl;
                // fvMatrix::setReference (closed domains: icoFoam.C:89, simpleFoam/pEqn.H:21) and setValues on copies of it
                FEqn.source() = src;
                FEqn.setReference(7, 0.5);
                std::vector<scalar> d = FEqn.diag().asHost(), so = FEqn.source().asHost();
                Info << "setReference cell 7: " << d[7] << " " << so[7] << std::endl;
                labelList cells{3, 11, 40}; scalarField vals{0.25, -0.5, 1.5};
                scalargpuField psiF(src);
                FEqn.setValues(labelgpuList(cells), scalargpuField(vals), psiF);
                scalar ss = 0, su = 0, sl = 0;
                for (scalar v : FEqn.source().asHost()) ss += v;
                for (scalar v : FEqn.upper().asHost()) su += v;
                for (scalar v : FEqn.lower().asHost()) sl += v;
                Info << "setValues sums source upper lower psi11: " << ss << " " << su << " " << sl << " " << psiF.asHost()[11] << std::endl;
            }
            scalargpuField psi(n);
            UEqn.relax(0.7, psi);
            UEqn.solve(psi, dictionary{{"solver", "PBiCG"}, {"preconditioner", "diagonal"}, {"tolerance", "1e-10"}, {"relTol", "0"}});
        }
        // ---- UEqn as a vector equation: fvMatrix<vector>::solveSegregated, per-component boundary coefficients ----
        {
            scalarField wts(nf, 1.0), phi(nf);
            for (label f = 0; f < nf; ++f) phi[f] = dir[f] == 0 ? 0.3 * h * h : 0.0;
            fvScalarMatrix conv("U", addr, patches, std::vector<bool>(6, false));
            fvm::div(conv, scalargpuField(wts), scalargpuField(phi));
            fvScalarMatrix tmpEqn("U", addr, patches, std::vector<bool>(6, false));   // ddt + div - laplacian, shared by the components
            fvm::ddt(tmpEqn, 1.0 / 1e-3, 1.0, scalargpuField(scalarField(n, h * h * h)), scalargpuField(n));
            tmpEqn += conv;
            tmpEqn -= pEqn;
            std::vector<scalar> cl = tmpEqn.lower().asHost(), cu = tmpEqn.upper().asHost(), cd = tmpEqn.diag().asHost();
            fvVectorMatrix UEqn("U", addr, patches, std::vector<bool>(6, false));
            UEqn.lower() = cl; UEqn.upper() = cu; UEqn.diag() = cd;
            for (int d = 0; d < 3; ++d) {
                scalarField s(n);
                for (label c = 0; c < n; ++c) s[c] = (2.0 * splitmixUniform(900 + d, c) - 1.0) * h * h * h;
                UEqn.source().component(d) = s;
                UEqn.internalCoeffs()[0].component(d) = scalarField(patches[0].size(), (2.0 + d) * h);   // differs per component
                UEqn.boundaryCoeffs()[1].component(d) = scalarField(patches[1].size(), 0.01 * (d + 1) * h * h * h);
            }
            vectorgpuField U(n);
            solverPerformance sp = UEqn.solve(U, dictionary{{"solver", "PBiCG"}, {"preconditioner", "DILU"}, {"tolerance", "1e-10"}, {"relTol", "0"}});
            Info << "solveSegregated max: " << sp.solverName() << " " << sp.initialResidual() << " " << sp.finalResidual() << " " << sp.nIterations() << std::endl;
        }
        // ---- error behaviour: unknown run-time name lists the valid ones (lduMatrixSolver.C:84-100) ----
        try {
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "PCGG"}, {"preconditioner", "DIC"}});
        } catch (const Foam::error& e) { Info << e.what() << std::endl; }
        try {   // a preconditioner of the other symmetry's table (lduMatrixPreconditioner.C:83-101): DILU on a symmetric matrix
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "PCG"}, {"preconditioner", "DILU"}});
        } catch (const Foam::error& e) { Info << e.what() << std::endl; }
        try {   // interpolateCorrection aborts in the reference (GAMGSolverInterpolate.C:180 notImplemented): same outcome
            scalargpuField psi(n);
            pEqn.solve(psi, dictionary{{"solver", "GAMG"}, {"smoother", "GaussSeidel"}, {"agglomerator", "faceAreaPair"}, {"nCellsInCoarsestLevel", "10"},
                                       {"mergeLevels", "1"}, {"interpolateCorrection", "true"}});
        } catch (const Foam::error& e) { Info << e.what() << std::endl; }
        Info << "End" << std::endl;
        return 0;
    } catch (const Foam::error& e) { std::cerr << e.what() << std::endl; return 1; }
}
