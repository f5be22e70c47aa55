// scalarTransportFoam.C -- applications/solvers/basic/scalarTransportFoam on the engine:
//
//     while (simple.loop())  while (simple.correctNonOrthogonal())
//         solve(fvm::ddt(T) + fvm::div(phi, T) - fvm::laplacian(DT, T) == fvOptions(T));        // (no fvOptions here)
//
// in a given velocity field, on a case directory: constant/polyMesh, constant/transportProperties (DT), system/controlDict (deltaT, endTime,
// writeFormat, writePrecision), system/fvSchemes (div(phi,T): Gauss linear | upwind | limitedLinear k; laplacian: Gauss linear corrected |
// uncorrected | orthogonal), system/fvSolution (solvers.T, SIMPLE.nNonOrthogonalCorrectors), 0/T and 0/U with fixedValue (inflow) and
// zeroGradient (outflow, walls) patches [U: also noSlip].  The whole equation is ONE assembly pass (fvm::assemble -> mi_fvm_assemble); the patch
// coefficients are the reference's: gaussConvectionScheme.C:96-110 (internalCoeffs = phi_b valueInternalCoeffs, boundaryCoeffs = -phi_b
// valueBoundaryCoeffs: fixedValue 0 / T_b, zeroGradient 1 / 0) and gaussLaplacianScheme.C:60-88 (fixedValue: DT |Sf| deltaCoeffs on both, with the
// sign of `- fvm::laplacian`; zeroGradient: none).  T goes back into the case at the end.   usage: scalarTransportFoam <caseDir> [-nSteps N]
// tests/test_scalartransportfoam.py walks the same statements on the oracle.
#include "polyMesh.H"
#include "solution.H"

#include <cmath>
#include <iomanip>
#include <memory>

using namespace Foam;

namespace
{
scalar lastNumber(const word& v) { const std::size_t at = v.find_last_of(' '); return std::strtod(v.c_str() + (at == std::string::npos ? 0 : at + 1), nullptr); }
scalargpuField product(const scalargpuField& x, const scalargpuField& y)
{
    scalargpuField out(x.size());
    fieldSubMul(out, x, y);
    fieldAxpby(out, -1.0, out, 0.0, out);
    return out;
}
}

int main(int argc, char** argv)
{
    try {
        if (argc < 2) { std::cerr << "usage: scalarTransportFoam <caseDir> [-nSteps N]" << std::endl; return 2; }
        const std::string caseDir = argv[1];
        Info << std::setprecision(12);
        polyMesh mesh(caseDir);
        const label n = mesh.nCells, nI = mesh.nInternalFaces(), nP = (label)mesh.boundary.size();
        const std::shared_ptr<dictTree> transport = readDictionaryFile(caseDir + "/constant/transportProperties");
        const std::shared_ptr<dictTree> control = readDictionaryFile(caseDir + "/system/controlDict");
        const scalar DT = lastNumber(transport->lookup("DT"));
        const scalar deltaT = lastNumber(control->lookup("deltaT")), endTime = lastNumber(control->lookup("endTime"));
        label nSteps = (label)std::llround(endTime / deltaT);
        for (int k = 2; k + 1 < argc; ++k) if (std::string(argv[k]) == "-nSteps") nSteps = (label)std::atoi(argv[k + 1]);
        const bool writeBinary = control->found("writeFormat") && control->lookup("writeFormat") == "binary";
        const int writePrecision = control->found("writePrecision") ? (int)lastNumber(control->lookup("writePrecision")) : 6;
        const solution fvSolution(caseDir);
        const fvSchemes schemes(caseDir);
        const label nNonOrthCorr = fvSolution.solutionDict().found("SIMPLE") ? fvSolution.dict("SIMPLE").lookupOrDefault<label>("nNonOrthogonalCorrectors", 0) : 0;
        if (schemes.ddtScheme("ddt(T)") != wordList{"Euler"}) FatalErrorIn("scalarTransportFoam", "ddtSchemes: only Euler");
        const wordList divT = schemes.divScheme("div(phi,T)");
        if (divT.size() < 2 || divT[0] != "Gauss" || (divT[1] != "linear" && divT[1] != "upwind" && !(divT[1] == "limitedLinear" && divT.size() == 3)))
            FatalErrorIn("scalarTransportFoam", "div(phi,T): Gauss linear | Gauss upwind | Gauss limitedLinear k");
        const bool upwind = divT[1] == "upwind", limited = divT[1] == "limitedLinear";
        const scalar limiterK = limited ? std::strtod(divT[2].c_str(), nullptr) : 0.0;
        const wordList lap = schemes.laplacianScheme("laplacian(DT,T)");
        if (lap.size() != 3 || lap[0] != "Gauss" || lap[1] != "linear" || (lap[2] != "orthogonal" && lap[2] != "uncorrected" && lap[2] != "corrected"))
            FatalErrorIn("scalarTransportFoam", "laplacianSchemes laplacian(DT,T): Gauss linear corrected | uncorrected | orthogonal");
        const bool corrected = lap[2] == "corrected";
        Info << "Create mesh: nCells " << n << " nInternalFaces " << nI << " patches " << nP << "; DT " << DT << " deltaT " << deltaT << " steps " << nSteps
             << " div(phi,T) " << divT[1] << " laplacian " << lap[2] << " nNonOrthogonalCorrectors " << nNonOrthCorr << std::endl;

        labelList lower(mesh.owner.begin(), mesh.owner.begin() + nI);
        std::vector<labelList> patchCells;
        for (label p = 0; p < nP; ++p) patchCells.push_back(mesh.patchFaceCells(p));
        lduAddressing addr(n, lower, mesh.neighbour);
        auto comp = [&](const vectorField& v, std::size_t b, std::size_t e) {
            vectorgpuField out((label)(e - b));
            for (direction d = 0; d < 3; ++d) { scalarField h(e - b); for (std::size_t i = b; i < e; ++i) h[i - b] = v[i][d]; out.component(d) = h; }
            return out;
        };
        const vectorgpuField SfI = comp(mesh.Sf, 0, (std::size_t)nI), Cc = comp(mesh.C, 0, (std::size_t)n);
        const vectorgpuField corrVecs = comp(mesh.nonOrthCorrectionVectors, 0, (std::size_t)nI);
        const scalargpuField V(mesh.V), weights(mesh.weights), deltaCoeffs(mesh.nonOrthDeltaCoeffs);
        scalarField dtMagSfH((std::size_t)nI); for (label f = 0; f < nI; ++f) dtMagSfH[f] = DT * mesh.magSf[f];
        const scalargpuField dtMagSf(dtMagSfH);
        scalargpuField negDtMagSf(nI); fieldAxpby(negDtMagSf, -1.0, dtMagSf, 0.0, dtMagSf);
        std::vector<std::unique_ptr<fvPatchCells>> patch;
        std::vector<vectorgpuField> patchSf;
        for (label p = 0; p < nP; ++p) {
            patch.emplace_back(new fvPatchCells(n, patchCells[p]));
            patchSf.push_back(comp(mesh.Sf, (std::size_t)mesh.boundary[p].startFace, (std::size_t)(mesh.boundary[p].startFace + mesh.boundary[p].nFaces)));
        }

        // ---- the velocity field and its face flux phi = linearInterpolate(U) & Sf (createPhi.H); boundary faces: U_b & Sf_b
        const vectorField U0 = readVolVectorInternalField(caseDir + "/0/U", n);
        const std::vector<patchFieldIn> Ub = readVolFieldBoundary(caseDir + "/0/U", 3), Tb = readVolFieldBoundary(caseDir + "/0/T", 1);
        if ((label)Ub.size() != nP || (label)Tb.size() != nP) FatalErrorIn("scalarTransportFoam", "0/U and 0/T need one boundaryField entry per patch");
        vectorgpuField U(n);
        for (direction d = 0; d < 3; ++d) { scalarField h((std::size_t)n); for (label c = 0; c < n; ++c) h[c] = U0[c][d]; U.component(d) = h; }
        scalargpuField phi(nI);
        fvc::fluxDiv(phi, nullptr, addr, weights, SfI, U);
        std::vector<scalargpuField> phiB, diffB, TbDev, icDev, bcDev;      // per patch: boundary flux, DT |Sf| deltaCoeffs, T_b (fixedValue), the matrix coefficients
        std::vector<bool> fixedT((std::size_t)nP, false);
        for (label p = 0; p < nP; ++p) {
            const label np = mesh.boundary[p].nFaces, f0 = mesh.boundary[p].startFace;
            if (Ub[p].patchName != mesh.boundary[p].name || Tb[p].patchName != mesh.boundary[p].name) FatalErrorIn("scalarTransportFoam", "boundaryField entries must follow the mesh's patch order");
            scalarField ph((std::size_t)np), df((std::size_t)np), tb((std::size_t)np, 0.0);
            for (label i = 0; i < np; ++i) {
                vector ub{0, 0, 0};
                if (Ub[p].type == "fixedValue") for (int d = 0; d < 3; ++d) ub[d] = Ub[p].uniform ? Ub[p].value[d] : Ub[p].value[3 * i + d];
                else if (Ub[p].type == "zeroGradient") ub = U0[(std::size_t)patchCells[p][i]];
                else if (Ub[p].type != "noSlip") FatalErrorIn("scalarTransportFoam", "U: fixedValue | zeroGradient | noSlip patches (patch " + Ub[p].patchName + " is " + Ub[p].type + ")");
                ph[i] = ub[0] * mesh.Sf[f0 + i][0] + ub[1] * mesh.Sf[f0 + i][1] + ub[2] * mesh.Sf[f0 + i][2];
                df[i] = DT * mesh.patchMagSf[p][i] * mesh.patchDeltaCoeffs[p][i];
            }
            if (Tb[p].type == "fixedValue") {
                if (!Tb[p].hasValue) FatalErrorIn("scalarTransportFoam", "T: fixedValue patch " + Tb[p].patchName + " without a value");
                fixedT[p] = true;
                for (label i = 0; i < np; ++i) tb[i] = Tb[p].uniform ? Tb[p].value[0] : Tb[p].value[i];
            } else if (Tb[p].type != "zeroGradient") FatalErrorIn("scalarTransportFoam", "T: fixedValue | zeroGradient patches (patch " + Tb[p].patchName + " is " + Tb[p].type + ")");
            phiB.emplace_back(ph); diffB.emplace_back(df); TbDev.emplace_back(tb);
            if (fixedT[p]) {                                   // convection 0 / -phi_b T_b; diffusion DT |Sf| deltaCoeffs / DT |Sf| deltaCoeffs T_b
                icDev.emplace_back(df);
                scalargpuField bc = product(diffB[p], TbDev[p]);
                fieldSubMul(bc, phiB[p], TbDev[p]);
                bcDev.push_back(bc);
            } else {                                           // zeroGradient: convection phi_b / 0; no diffusive flux
                icDev.emplace_back(ph);
                bcDev.emplace_back(scalarField((std::size_t)np, 0.0));
            }
        }
        scalargpuField T(readVolScalarInternalField(caseDir + "/0/T", n));
        auto gaussGrad = [&](vectorgpuField& g, const scalargpuField& vf) {       // fvc::grad(T), Gauss linear, with T_b on fixedValue patches / patchInternalField
            scalargpuField ff(nI);
            fvc::interpolate(ff, addr, weights, vf);
            miCheck(mi_gauss_grad(addr.handle(), SfI.component(0).data(), SfI.component(1).data(), SfI.component(2).data(), ff.data(), nullptr,
                                  g.component(0).data(), g.component(1).data(), g.component(2).data()), "gaussGrad::gradf");
            for (label q = 0; q < nP; ++q) {
                scalargpuField pif(patch[q]->size());
                const scalargpuField* pv = &TbDev[q];
                if (!fixedT[q]) { patch[q]->patchInternalField(vf, pif); pv = &pif; }
                for (direction d = 0; d < 3; ++d) patch[q]->addProduct(patchSf[q].component(d), *pv, g.component(d));
            }
            for (direction d = 0; d < 3; ++d) fieldDivide(g.component(d), g.component(d), V);
        };
        const dictionary TControls = fvSolution.solverDict("T");
        const scalar rDeltaT = 1.0 / deltaT;
        const std::vector<bool> notCoupled((std::size_t)nP, false);
        scalargpuField w(nI);
        if (upwind) upwindWeights(w, phi);                      // (phi does not change: the upwind weights are formed once)

        Info << std::endl << "Calculating scalar transport" << std::endl << std::endl;
        for (label step = 1; step <= nSteps; ++step) {
            Info << "Time = " << step * deltaT << std::endl << std::endl;
            const scalargpuField Told(T);
            for (label nonOrth = 0; nonOrth <= nNonOrthCorr; ++nonOrth) {
                vectorgpuField gT(n);
                if (limited || corrected) gaussGrad(gT, T);
                if (limited) limitedLinearWeights(w, addr, limiterK, weights, phi, T, gT, Cc);
                fvScalarMatrix TEqn("T", addr, patchCells, notCoupled);
                fvm::assemble(TEqn, rDeltaT, 1.0, V, Told, &phi, (upwind || limited) ? &w : &weights, &deltaCoeffs, &dtMagSf);
                for (label q = 0; q < nP; ++q) { TEqn.internalCoeffs()[q] = icDev[q]; TEqn.boundaryCoeffs()[q] = bcDev[q]; }
                if (corrected) {                                // - fvm::laplacian(DT, T), corrected: source += V*div(DT |Sf| correction(T))
                    scalargpuField cf(nI), d(n);
                    fvc::snGradCorrectionFlux(cf, addr, corrVecs, weights, gT, negDtMagSf);
                    fvc::surfaceIntegrate(d, addr, cf, &V);
                    fieldSubMul(TEqn.source(), V, d);
                }
                TEqn.solve(T, TControls);
            }
            Info << std::endl;
        }
        std::ostringstream tn; tn << std::setprecision(10) << nSteps * deltaT;
        std::vector<patchFieldOut> bt;
        for (label q = 0; q < nP; ++q) {
            patchFieldOut e; e.patchName = mesh.boundary[q].name; e.type = Tb[q].type; e.hasValue = fixedT[q];
            if (e.hasValue) e.value = TbDev[q].asHost();
            bt.push_back(e);
        }
        writeVolScalarField(caseDir, tn.str(), "T", "[0 0 0 1 0 0 0]", T.asHost(), bt, writeBinary, writePrecision);
        Info << "wrote " << caseDir << "/" << tn.str() << "/T" << std::endl << "End" << std::endl;
        return 0;
    } catch (const Foam::error& e) { std::cerr << e.what() << std::endl; return 1; }
}
