// icoFoam.C -- BASELINE config 1's application on the engine: transient incompressible laminar flow, PISO, written against the mirror
// the way applications/solvers/incompressible/icoFoam/icoFoam.C is written against OpenFOAM -- statement for statement:
//
//     fvVectorMatrix UEqn(fvm::ddt(U) + fvm::div(phi, U) - fvm::laplacian(nu, U));
//     solve(UEqn == -fvc::grad(p));
//     PISO: rAU = 1/UEqn.A(); HbyA = rAU*UEqn.H(); phiHbyA = (interpolate(HbyA) & Sf) + interpolate(rAU)*ddtCorr(U, phi);
//           pEqn(fvm::laplacian(rAU, p) == fvc::div(phiHbyA)); setReference; solve; phi = phiHbyA - pEqn.flux();
//           continuity errors; U = HbyA - rAU*fvc::grad(p)
//
// on a real case directory: constant/polyMesh, constant/transportProperties (nu), system/controlDict (deltaT, endTime, writeFormat,
// writePrecision), system/fvSchemes (div(phi,U): Gauss linear | Gauss upwind | Gauss limitedLinear k; laplacian: Gauss linear corrected |
// uncorrected -- the corrected form with its explicit non-orthogonal flux in UEqn's and pEqn's sources and in pEqn.flux()), system/fvSolution (solvers U, p [pFinal]; PISO: nCorrectors,
// pRefCell, pRefValue), 0/U (fixedValue / noSlip patches), 0/p (zeroGradient patches: a closed domain, hence the reference level).
// UEqn is the mirror's fvVectorMatrix: one set of coefficients and three sources out of ONE assembly pass (mi_fvm_assemble), the three
// components solved as one batched PBiCG (fvMatrix<vector>::solveSegregated on the engine), A() and H() as fvMatrix.C:1374-1506; every
// field operation is a call of the path.
// At the end U and p go back into the case as <case>/<endTime>/{U,p}.   usage: icoFoam <caseDir> [-nSteps N]
// tests/test_icofoam.py runs the same steps on the oracle and compares every solver line, the continuity errors and the written fields.
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
    scalargpuField out(x.size());            // zero-initialised
    fieldSubMul(out, x, y);                  // -(x*y), the product rounded once
    fieldAxpby(out, -1.0, out, 0.0, out);    // sign flip: exact
    return out;
}
}

int main(int argc, char** argv)
{
    try {
        if (argc < 2) { std::cerr << "usage: icoFoam <caseDir> [-nSteps N]" << std::endl; return 2; }
        const std::string caseDir = argv[1];
        Info << std::setprecision(12);
        polyMesh mesh(caseDir);
        const label n = mesh.nCells, nI = mesh.nInternalFaces(), nP = (label)mesh.boundary.size();
        const std::shared_ptr<dictTree> transport = readDictionaryFile(caseDir + "/constant/transportProperties");
        const std::shared_ptr<dictTree> control = readDictionaryFile(caseDir + "/system/controlDict");
        const scalar nu = lastNumber(transport->lookup("nu"));                       // "nu [0 2 -1 0 0 0 0] 0.01" or "0.01"
        const scalar deltaT = lastNumber(control->lookup("deltaT")), endTime = lastNumber(control->lookup("endTime"));
        label nSteps = (label)std::llround(endTime / deltaT);
        for (int k = 2; k + 1 < argc; ++k) if (std::string(argv[k]) == "-nSteps") nSteps = (label)std::atoi(argv[k + 1]);
        const bool writeBinary = control->found("writeFormat") && control->lookup("writeFormat") == "binary";
        const int writePrecision = control->found("writePrecision") ? (int)lastNumber(control->lookup("writePrecision")) : 6;
        const solution fvSolution(caseDir);
        const fvSchemes schemes(caseDir);
        const dictionary piso = fvSolution.dict("PISO");
        const label nCorr = piso.lookupOrDefault<label>("nCorrectors", 1), nNonOrthCorr = piso.lookupOrDefault<label>("nNonOrthogonalCorrectors", 0);
        const label pRefCell = piso.lookupOrDefault<label>("pRefCell", 0); const scalar pRefValue = piso.lookupOrDefault<scalar>("pRefValue", 0.0);
        if (schemes.ddtScheme("ddt(U)") != wordList{"Euler"}) FatalErrorIn("icoFoam", "ddtSchemes: only Euler");
        const wordList divU = schemes.divScheme("div(phi,U)");
        if (divU.size() < 2 || divU[0] != "Gauss" || (divU[1] != "linear" && divU[1] != "upwind" && !(divU[1] == "limitedLinear" && divU.size() == 3)))
            FatalErrorIn("icoFoam", "div(phi,U): Gauss linear | Gauss upwind | Gauss limitedLinear k");
        bool corrected = false;
        for (const char* term : {"laplacian(nu,U)", "laplacian((1|A(U)),p)"}) {
            const wordList l = schemes.laplacianScheme(term);
            if (l.size() != 3 || l[0] != "Gauss" || l[1] != "linear" || (l[2] != "orthogonal" && l[2] != "uncorrected" && l[2] != "corrected"))
                FatalErrorIn("icoFoam", std::string("laplacianSchemes ") + term + ": Gauss linear corrected | uncorrected | orthogonal");
            corrected = corrected || l[2] == "corrected";
        }
        const bool upwind = divU[1] == "upwind", limited = divU[1] == "limitedLinear";
        const scalar limiterK = limited ? std::strtod(divU[2].c_str(), nullptr) : 0.0;
        Info << "Create mesh: nCells " << n << " nInternalFaces " << nI << " patches " << nP << "; nu " << nu << " deltaT " << deltaT << " steps " << nSteps
             << " nCorrectors " << nCorr << " div(phi,U) " << divU[1] << std::endl;

        // ---- mesh-side device fields
        labelList lower(mesh.owner.begin(), mesh.owner.begin() + nI);
        std::vector<labelList> patchCells;
        for (label p = 0; p < nP; ++p) patchCells.push_back(mesh.patchFaceCells(p));
        lduAddressing addr(n, lower, mesh.neighbour);
        auto comp = [&](const vectorField& v, std::size_t b, std::size_t e) {
            vectorgpuField out((label)(e - b));
            for (direction d = 0; d < 3; ++d) { scalarField h(e - b); for (std::size_t i = b; i < e; ++i) h[i - b] = v[i][d]; out.component(d) = h; }
            return out;
        };
        const vectorgpuField SfI = comp(mesh.Sf, 0, (std::size_t)nI);
        const scalargpuField V(mesh.V), weights(mesh.weights), deltaCoeffs(mesh.nonOrthDeltaCoeffs);
        scalarField nuMagSfH((std::size_t)nI); for (label f = 0; f < nI; ++f) nuMagSfH[f] = nu * mesh.magSf[f];
        const scalargpuField nuMagSf(nuMagSfH), magSfI(scalarField(mesh.magSf.begin(), mesh.magSf.begin() + nI));
        scalarField onesH((std::size_t)n, 1.0); const scalargpuField ones(onesH);
        std::vector<std::unique_ptr<fvPatchCells>> patch;
        std::vector<vectorgpuField> patchSf;
        for (label p = 0; p < nP; ++p) {
            patch.emplace_back(new fvPatchCells(n, patchCells[p]));
            patchSf.push_back(comp(mesh.Sf, (std::size_t)mesh.boundary[p].startFace, (std::size_t)(mesh.boundary[p].startFace + mesh.boundary[p].nFaces)));
        }
        const scalarField faceAreaPair = mesh.faceAreaPairWeights();
        setFaceAreaPairWeights(&faceAreaPair);

        // ---- fields: U (fixedValue | noSlip on every patch), p (zeroGradient on every patch), phi = interpolate(U) & Sf
        vectorField U0 = readVolVectorInternalField(caseDir + "/0/U", n);
        const std::vector<patchFieldIn> Ub = readVolFieldBoundary(caseDir + "/0/U", 3), pb = readVolFieldBoundary(caseDir + "/0/p", 1);
        if ((label)Ub.size() != nP || (label)pb.size() != nP) FatalErrorIn("icoFoam", "0/U and 0/p need one boundaryField entry per patch");
        std::vector<vectorField> UbVal((std::size_t)nP);            // boundary values of U per patch face
        for (label p = 0; p < nP; ++p) {
            const label np = mesh.boundary[p].nFaces;
            if (Ub[p].patchName != mesh.boundary[p].name || pb[p].patchName != mesh.boundary[p].name) FatalErrorIn("icoFoam", "boundaryField entries must follow the mesh's patch order");
            if (pb[p].type != "zeroGradient") FatalErrorIn("icoFoam", "p: zeroGradient patches only (patch " + pb[p].patchName + " is " + pb[p].type + ")");
            UbVal[p].assign((std::size_t)np, vector{0, 0, 0});
            if (Ub[p].type == "fixedValue") {
                if (!Ub[p].hasValue) FatalErrorIn("icoFoam", "U: fixedValue patch " + Ub[p].patchName + " without a value");
                for (label i = 0; i < np; ++i) for (int d = 0; d < 3; ++d) UbVal[p][i][d] = Ub[p].uniform ? Ub[p].value[d] : Ub[p].value[3 * i + d];
            } else if (Ub[p].type != "noSlip") FatalErrorIn("icoFoam", "U: fixedValue | noSlip patches only (patch " + Ub[p].patchName + " is " + Ub[p].type + ")");
        }
        vectorgpuField U(n), gradP(n), HbyA(n);
        for (direction d = 0; d < 3; ++d) { scalarField h((std::size_t)n); for (label c = 0; c < n; ++c) h[c] = U0[c][d]; U.component(d) = h; }
        scalargpuField p(readVolScalarInternalField(caseDir + "/0/p", n));
        scalargpuField phi(nI), phiHbyA(nI), ddtCorrF(nI), rAUf(nI), gammaMagSf(nI), divPhi(n), pflux(nI);
        fvc::fluxDiv(phi, nullptr, addr, weights, SfI, U);                                // createPhi.H: linearInterpolate(U) & mesh.Sf()
        // per patch: the boundary flux (U_b & Sf_b), the diffusive coefficient nu |Sf| deltaCoeffs, U_b per component (device)
        std::vector<scalargpuField> phiB, diffB; std::vector<vectorgpuField> UbDev;
        for (label p = 0; p < nP; ++p) {
            const label np = mesh.boundary[p].nFaces, f0 = mesh.boundary[p].startFace;
            scalarField ph((std::size_t)np), df((std::size_t)np);
            for (label i = 0; i < np; ++i) {
                ph[i] = UbVal[p][i][0] * mesh.Sf[f0 + i][0] + UbVal[p][i][1] * mesh.Sf[f0 + i][1] + UbVal[p][i][2] * mesh.Sf[f0 + i][2];
                df[i] = nu * mesh.patchMagSf[p][i] * mesh.patchDeltaCoeffs[p][i];
            }
            phiB.emplace_back(ph); diffB.emplace_back(df);
            UbDev.push_back(comp(UbVal[p], 0, (std::size_t)np));
        }
        const vectorgpuField Cc = comp(mesh.C, 0, (std::size_t)n);
        std::vector<scalargpuField> magSqrUb;                          // magSqr of the boundary values (the limiter's gradient takes them on the patch faces)
        for (label q = 0; q < nP; ++q) {
            scalarField h(UbVal[q].size());
            for (std::size_t i = 0; i < h.size(); ++i) h[i] = (UbVal[q][i][0] * UbVal[q][i][0] + UbVal[q][i][1] * UbVal[q][i][1]) + UbVal[q][i][2] * UbVal[q][i][2];
            magSqrUb.emplace_back(h);
        }
        // fvc::grad(vf), Gauss linear: internal faces, then vf_b Sf_b of every patch (patchValue q given: fixedValue; nullptr: zeroGradient = patchInternalField), / V
        auto gaussGrad = [&](vectorgpuField& g, const scalargpuField& vf, const std::vector<const scalargpuField*>& patchValue) {
            scalargpuField ff(nI);
            fvc::interpolate(ff, addr, weights, vf);
            miCheck(mi_gauss_grad(addr.handle(), SfI.component(0).data(), SfI.component(1).data(), SfI.component(2).data(), ff.data(), nullptr,
                                  g.component(0).data(), g.component(1).data(), g.component(2).data()), "gaussGrad::gradf");
            for (label q = 0; q < nP; ++q) {
                scalargpuField pif(patch[q]->size());
                const scalargpuField* pv = patchValue[(std::size_t)q];
                if (!pv) { patch[q]->patchInternalField(vf, pif); pv = &pif; }
                for (direction d = 0; d < 3; ++d) patch[q]->addProduct(patchSf[q].component(d), *pv, g.component(d));
            }
            for (direction d = 0; d < 3; ++d) fieldDivide(g.component(d), g.component(d), V);
        };
        const std::vector<const scalargpuField*> zeroGradientPatches((std::size_t)nP, nullptr);
        auto gradOfP = [&]() { gaussGrad(gradP, p, zeroGradientPatches); };
        const vectorgpuField corrVecs = comp(mesh.nonOrthCorrectionVectors, 0, (std::size_t)nI);
        scalargpuField negNuMagSf(nI); fieldAxpby(negNuMagSf, -1.0, nuMagSf, 0.0, nuMagSf);
        // source -= V * fvc::div(gammaMagSf * correction(vf)): the explicit part of a `corrected` fvm::laplacian(gamma, vf) (gaussLaplacianSchemes.C:64-90);
        // returns the correction flux (pEqn keeps it: fvMatrix::flux() adds faceFluxCorrectionPtr, fvMatrix.C:1655-1658)
        auto correctLaplacian = [&](fvScalarMatrix& M, const vectorgpuField& gradVf, const scalargpuField& gMagSf, scalargpuField& corrFlux) {
            fvc::snGradCorrectionFlux(corrFlux, addr, corrVecs, weights, gradVf, gMagSf);
            scalargpuField d(n);
            fvc::surfaceIntegrate(d, addr, corrFlux, &V);
            fieldSubMul(M.source(), V, d);
        };
        const dictionary UControls = fvSolution.solverDict("U"), pControls = fvSolution.solverDict("p");
        const dictionary pFinalControls = fvSolution.solutionDict().subDict("solvers").found("pFinal") ? fvSolution.solverDict("pFinal") : pControls;
        const scalar rDeltaT = 1.0 / deltaT;
        scalar cumulativeContErr = 0, totalV = 0;
        for (scalar v : mesh.V) totalV += v;
        const std::vector<bool> notCoupled((std::size_t)nP, false);

        Info << std::endl << "Starting time loop" << std::endl << std::endl;
        for (label step = 1; step <= nSteps; ++step) {
            Info << "Time = " << step * deltaT << std::endl << std::endl;
            const vectorgpuField Uold(U); const scalargpuField phiOld(phi);
            scalargpuField upw(nI);
            if (upwind) upwindWeights(upw, phi);
            if (limited) {
                // limitedLinear on a vector field: LimitedScheme<vector, limitedLinearLimiter<NVDTVD>, limitFuncs::magSqr> (limitedLinear.C) -- ONE limiter for
                // the three components, formed from magSqr(U) and its Gauss gradient (LimitedScheme.C:39-110, NVDTVD.H); magSqr = (Ux Ux + Uy Uy) + Uz Uz,
                // every product and sum rounded on its own here
                scalargpuField m2 = product(U.component(0), U.component(0));
                fieldAxpby(m2, 1.0, m2, 1.0, product(U.component(1), U.component(1)));
                fieldAxpby(m2, 1.0, m2, 1.0, product(U.component(2), U.component(2)));
                scalargpuField m2f(nI);
                fvc::interpolate(m2f, addr, weights, m2);
                vectorgpuField g(n);
                miCheck(mi_gauss_grad(addr.handle(), SfI.component(0).data(), SfI.component(1).data(), SfI.component(2).data(), m2f.data(), nullptr,
                                      g.component(0).data(), g.component(1).data(), g.component(2).data()), "gaussGrad::gradf");
                for (label q = 0; q < nP; ++q) for (direction d = 0; d < 3; ++d) patch[q]->addProduct(patchSf[q].component(d), magSqrUb[q], g.component(d));
                for (direction d = 0; d < 3; ++d) fieldDivide(g.component(d), g.component(d), V);
                limitedLinearWeights(upw, addr, limiterK, weights, phi, m2, g, Cc);
            }
            const scalargpuField& convWeights = (upwind || limited) ? upw : weights;
            // fvVectorMatrix UEqn(fvm::ddt(U) + fvm::div(phi, U) - fvm::laplacian(nu, U)): the coefficients once, the three sources out of the same pass
            fvVectorMatrix UEqn("U", addr, patchCells, notCoupled);
            gradOfP();
            fvm::assemble(UEqn, rDeltaT, 1.0, V, Uold, &phi, &convWeights, &deltaCoeffs, &nuMagSf);
            for (label q = 0; q < nP; ++q)
                for (direction d = 0; d < 3; ++d) {
                    // fixedValue patch: convection valueInternalCoeffs 0 / valueBoundaryCoeffs U_b (gaussConvectionScheme.C:96-110: boundaryCoeffs = -phi_b U_b);
                    // diffusion, with the sign of `- fvm::laplacian`: internalCoeffs = nu |Sf| deltaCoeffs, boundaryCoeffs = nu |Sf| deltaCoeffs U_b
                    UEqn.internalCoeffs()[q].component(d) = diffB[q];
                    scalargpuField bc = product(diffB[q], UbDev[q].component(d));
                    fieldSubMul(bc, phiB[q], UbDev[q].component(d));
                    UEqn.boundaryCoeffs()[q].component(d) = bc;
                }
            if (corrected)   // - fvm::laplacian(nu, U), corrected: its explicit part enters with the opposite sign -- gamma = -nu
                for (direction d = 0; d < 3; ++d) {
                    std::vector<const scalargpuField*> fixedValues;
                    for (label q = 0; q < nP; ++q) fixedValues.push_back(&UbDev[q].component(d));
                    vectorgpuField gU(n); scalargpuField cf(nI), dv(n);
                    gaussGrad(gU, Uold.component(d), fixedValues);
                    fvc::snGradCorrectionFlux(cf, addr, corrVecs, weights, gU, negNuMagSf);
                    fvc::surfaceIntegrate(dv, addr, cf, &V);
                    fieldSubMul(UEqn.source().component(d), V, dv);
                }
            {   // solve(UEqn == -fvc::grad(p)): the temporary's source is source - V*grad(p)   (fvMatrix.C: operator==, operator-(fvMatrix, field));
                // fvMatrix<vector>::solveSegregated -- the three components as ONE batched PBiCG on the engine (mi_pbicg_solve_multi)
                const vectorgpuField keep(UEqn.source());
                for (direction d = 0; d < 3; ++d) fieldSubMul(UEqn.source().component(d), V, gradP.component(d));
                UEqn.solve(U, UControls);
                for (direction d = 0; d < 3; ++d) UEqn.source().component(d) = keep.component(d);
            }
            // --- PISO loop
            for (label corr = 0; corr < nCorr; ++corr) {
                scalargpuField A(n), rAU(n);
                UEqn.A(A, V);
                fieldDivide(rAU, ones, A);                                           // volScalarField rAU(1.0/UEqn.A());
                {                                                                    // HbyA = rAU*UEqn.H();
                    vectorgpuField H(n);
                    UEqn.H(H, U, V);
                    for (direction d = 0; d < 3; ++d) HbyA.component(d) = product(rAU, H.component(d));
                }
                fvc::interpolate(rAUf, addr, weights, rAU);
                fvc::ddtCorr(ddtCorrF, addr, rDeltaT, weights, SfI, Uold, phiOld);
                // phiHbyA = (fvc::interpolate(HbyA) & mesh.Sf()) + fvc::interpolate(rAU)*fvc::ddtCorr(U, phi), and sum over the cell's faces = V*fvc::div(phiHbyA)
                fvc::fluxDiv(phiHbyA, &divPhi, addr, weights, SfI, HbyA, &rAUf, &ddtCorrF, nullptr);
                for (label q = 0; q < nP; ++q) patch[q]->add(phiB[q], divPhi);      // boundary faces: HbyA_b = U_b
                // adjustPhi(phiHbyA, U, p): no inflow / outflow on fixedValue walls whose flux is zero -- nothing to adjust in a closed domain
                for (label nonOrth = 0; nonOrth <= nNonOrthCorr; ++nonOrth) {
                    fvScalarMatrix pEqn("p", addr, patchCells, notCoupled);
                    gammaMagSf = product(rAUf, magSfI);
                    fvm::laplacian(pEqn, deltaCoeffs, gammaMagSf);                   // fvm::laplacian(rAU, p) == fvc::div(phiHbyA)
                    pEqn.source() = divPhi;
                    scalargpuField pCorrFlux(nI);
                    if (corrected) { gradOfP(); correctLaplacian(pEqn, gradP, gammaMagSf, pCorrFlux); }
                    pEqn.setReference(pRefCell, pRefValue);
                    pEqn.solve(p, (corr == nCorr - 1 && nonOrth == nNonOrthCorr) ? pFinalControls : pControls);
                    if (nonOrth == nNonOrthCorr) {                                   // phi = phiHbyA - pEqn.flux();
                        FieldFieldScalar bflux;
                        pEqn.flux(pflux, bflux, p);
                        if (corrected) fieldAxpby(pflux, 1.0, pflux, 1.0, pCorrFlux);   // + faceFluxCorrection (formed from p before this solve)
                        fieldAxpby(phi, 1.0, phiHbyA, -1.0, pflux);
                    }
                }
                {   // continuityErrs.H
                    scalargpuField contErr(n);
                    fvc::surfaceIntegrate(contErr, addr, phi, nullptr);
                    for (label q = 0; q < nP; ++q) patch[q]->add(phiB[q], contErr);
                    const scalarField ce = contErr.asHost();                         // = V * fvc::div(phi)
                    scalar sumLocal = 0, global = 0;
                    for (label c = 0; c < n; ++c) { sumLocal += std::fabs(ce[c]); global += ce[c]; }
                    sumLocal *= deltaT / totalV; global *= deltaT / totalV; cumulativeContErr += global;
                    Info << "time step continuity errors : sum local = " << sumLocal << ", global = " << global << ", cumulative = " << cumulativeContErr << std::endl;
                }
                gradOfP();
                for (direction d = 0; d < 3; ++d) { U.component(d) = HbyA.component(d); fieldSubMul(U.component(d), rAU, gradP.component(d)); }   // U = HbyA - rAU*fvc::grad(p)
            }
            Info << std::endl;
        }
        // runTime.write(): U and p of the last time
        std::ostringstream tn; tn << std::setprecision(10) << nSteps * deltaT;
        {
            vectorField Uh((std::size_t)n);
            for (direction d = 0; d < 3; ++d) { const scalarField h = U.component(d).asHost(); for (label c = 0; c < n; ++c) Uh[c][d] = h[c]; }
            std::vector<patchFieldOut> bu, bp;
            for (label q = 0; q < nP; ++q) {
                patchFieldOut e; e.patchName = mesh.boundary[q].name; e.type = Ub[q].type; e.hasValue = Ub[q].type == "fixedValue";
                if (e.hasValue) for (const vector& v : UbVal[q]) e.value.insert(e.value.end(), v.begin(), v.end());
                bu.push_back(e);
                patchFieldOut z; z.patchName = mesh.boundary[q].name; z.type = "zeroGradient"; bp.push_back(z);
            }
            writeVolVectorField(caseDir, tn.str(), "U", "[0 1 -1 0 0 0 0]", Uh, bu, writeBinary, writePrecision);
            writeVolScalarField(caseDir, tn.str(), "p", "[0 2 -2 0 0 0 0]", p.asHost(), bp, writeBinary, writePrecision);
            Info << "wrote " << caseDir << "/" << tn.str() << "/{U,p}" << std::endl;
        }
        Info << "End" << std::endl;
        return 0;
    } catch (const Foam::error& e) { std::cerr << e.what() << std::endl; return 1; }
}
