/* c_abi_demo.c -- the drop-in boundary used from plain C99: no C++, no Python, no torch.
 * Builds the LDU addressing and a symmetric pressure-like matrix of an nx x ny x nz box on the host, hands them to the engine
 * through include/mi_ldu.h, solves with PCG + DIC (= AINV in RapidCFD, DICPreconditioner.C:42-58) and prints the
 * solverPerformance the way OpenFOAM does.  tests/test_c_abi_demo.py compiles it with gcc and checks the line against the oracle.
 *
 *   gcc -std=c99 -O2 examples/c_abi_demo.c -Iinclude -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ \
 *       -Lrapidcfd-dev_amd -lrapidcfd_amd -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/rapidcfd-dev_amd -o c_abi_demo
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "mi_ldu.h"

#define CK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, mi_last_error()); return 1; } } while (0)
#define HK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int nx = argc > 1 ? atoi(argv[1]) : 24, ny = argc > 2 ? atoi(argv[2]) : 20, nz = argc > 3 ? atoi(argv[3]) : 16;
    const int32_t n = nx * ny * nz;
    const int32_t nf = 3 * n - (ny * nz + nx * nz + nx * ny);
    int32_t *lower = malloc(sizeof(int32_t) * (size_t)nf), *upper = malloc(sizeof(int32_t) * (size_t)nf);
    double *up = malloc(sizeof(double) * (size_t)nf), *diag = calloc((size_t)n, sizeof(double)), *src = malloc(sizeof(double) * (size_t)n);
    int32_t f = 0;
    /* OpenFOAM face order: for each cell ascending, its +x, +y, +z neighbours (owner-sorted, upper-triangular) */
    for (int k = 0; k < nz; k++) for (int j = 0; j < ny; j++) for (int i = 0; i < nx; i++) {
        const int32_t c = i + nx * (j + ny * k);
        if (i + 1 < nx) { lower[f] = c; upper[f] = c + 1; f++; }
        if (j + 1 < ny) { lower[f] = c; upper[f] = c + nx; f++; }
        if (k + 1 < nz) { lower[f] = c; upper[f] = c + nx * ny; f++; }
    }
    for (f = 0; f < nf; f++) { up[f] = -(1.0 + 0.001 * (double)(f % 97)); diag[lower[f]] -= up[f]; diag[upper[f]] -= up[f]; }
    for (int32_t c = 0; c < n; c++) {
        diag[c] += 0.05 + 0.001 * (double)(c % 13);
        src[c] = (double)(((uint32_t)c * 2654435761u) % 1000u) / 1000.0 - 0.5;
    }

    if (!mi_device_available()) { fprintf(stderr, "no gfx950 device: %s\n", mi_last_error()); return 2; }
    mi_ctx_t ctx; mi_addr_t addr; mi_matrix_t A;
    CK(mi_ctx_create(0, NULL, &ctx));
    CK(mi_addr_create(ctx, n, nf, lower, upper, 0, NULL, NULL, &addr));
    CK(mi_matrix_create(addr, &A));
    double *dDiag, *dUp, *dSrc, *dPsi, *dRes;
    HK(hipMalloc((void **)&dDiag, sizeof(double) * (size_t)n)); HK(hipMalloc((void **)&dUp, sizeof(double) * (size_t)nf));
    HK(hipMalloc((void **)&dSrc, sizeof(double) * (size_t)n)); HK(hipMalloc((void **)&dPsi, sizeof(double) * (size_t)n));
    HK(hipMalloc((void **)&dRes, sizeof(double) * (size_t)n));
    HK(hipMemcpy(dDiag, diag, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    HK(hipMemcpy(dUp, up, sizeof(double) * (size_t)nf, hipMemcpyHostToDevice));
    HK(hipMemcpy(dSrc, src, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    HK(hipMemset(dPsi, 0, sizeof(double) * (size_t)n));
    CK(mi_matrix_set_coeffs(A, dDiag, dUp, NULL));                          /* NULL lower: symmetric, as lduMatrix stores it */

    mi_solver_controls ctl; ctl.tolerance = 1e-9; ctl.relTol = 0.0; ctl.maxIter = 1000; ctl.minIter = 0;
    mi_solver_perf perf;
    CK(mi_pcg_solve(A, dPsi, dSrc, &ctl, MI_PRECOND_AINV, &perf, NULL, 0));
    printf("AINVPCG:  Solving for p, Initial residual = %.17g, Final residual = %.17g, No Iterations %d\n",
           perf.initialResidual, perf.finalResidual, (int)perf.nIterations);
    /* independent check through two more entry points: |b - A psi|_1 / normFactor */
    double sumMag = 0.0;
    CK(mi_residual(A, dPsi, dSrc, dRes));
    CK(mi_sum_mag(ctx, dRes, n, &sumMag));
    printf("check: sum|b - A psi| / normFactor = %.17g\n", sumMag / perf.normFactor);
    CK(mi_matrix_destroy(A)); CK(mi_addr_destroy(addr)); CK(mi_ctx_destroy(ctx));
    hipFree(dDiag); hipFree(dUp); hipFree(dSrc); hipFree(dPsi); hipFree(dRes);
    free(lower); free(upper); free(up); free(diag); free(src);
    printf("End\n");
    return 0;
}
