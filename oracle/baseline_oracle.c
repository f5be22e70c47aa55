/*
 * baseline_oracle.c -- the CPU BASELINE leg of bench.py (test infrastructure, NOT product code; see ldu_oracle.c).
 *
 * Upstream OpenFOAM's CPU lduMatrix path as it runs on a node: one MPI rank per core over a slab decomposition, no
 * threading inside a rank.  Here every domain of the orc_system is driven by its own OpenMP thread,
 * which plays the rank: face-loop Amul (OpenFOAM-2.3.x lduMatrixATmul.C: Apsi = D psi; for every face
 * Apsi[u] += lower*psi[l]; Apsi[l] += upper*psi[u] -- restated in ldu_oracle.c orc_amul_faceloop), processor-patch
 * update reading the neighbour domain's cells (shared memory stands in for MPI), diagonal-preconditioned PCG
 * (PCG.C:68-208 loop structure), the three global sums per iteration combined in rank order.  Fixed iteration count,
 * plain double sums: this is a TIMING kernel, parity is checked elsewhere.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>
#include "ldu_oracle.h"

static void dom_amul_faceloop(const orc_system *s, int d, const scalar *psiAll, scalar *yAll)
{
    const orc_domain *m = &s->dom[d];
    const scalar *x = psiAll + m->offset; scalar *y = yAll + m->offset;
    for (label c = 0; c < m->nCells; c++) y[c] = m->diag[c] * x[c];
    for (label f = 0; f < m->nFaces; f++) {
        y[m->upper[f]] += m->lowerC[f] * x[m->lower[f]];
        y[m->lower[f]] += m->upperC[f] * x[m->upper[f]];
    }
    for (int p = 0; p < m->nIfaces; p++) {
        const orc_iface *me = &m->ifaces[p];
        const orc_domain *nb = &s->dom[me->nbrDomain];
        const orc_iface *ot = &nb->ifaces[me->nbrPatch];
        const scalar *xn = psiAll + nb->offset;
        for (label i = 0; i < me->nFaces; i++) y[me->faceCells[i]] -= me->bouCoeffs[i] * xn[ot->faceCells[i]];
    }
}

/* returns the number of iterations done; *seconds = wall time of the iteration loop; *residual = sum|rA| at the end */
int orc_baseline_pcg(const orc_system *s, const scalar *source, int nIter, double *seconds, double *residual)
{
    const int D = s->nDomains;
    const int64_t n = s->nTotal;
    scalar *psi = (scalar *)calloc((size_t)n, sizeof(scalar)), *pA = (scalar *)calloc((size_t)n, sizeof(scalar));
    scalar *wA = (scalar *)calloc((size_t)n, sizeof(scalar)), *rA = (scalar *)malloc(sizeof(scalar) * (size_t)n);
    scalar *rD = (scalar *)malloc(sizeof(scalar) * (size_t)n);
    scalar *part = (scalar *)calloc((size_t)D * 8, sizeof(scalar)); /* one cache line per rank */
    memcpy(rA, source, sizeof(scalar) * (size_t)n);                /* psi0 = 0 */
    double t0 = 0, t1 = 0;
    #pragma omp parallel num_threads(D)
    {
        const int d = omp_get_thread_num();
        const orc_domain *m = &s->dom[d];
        const int64_t o = m->offset;
        for (label i = 0; i < m->nCells; i++) rD[o + i] = 1.0 / m->diag[i];
        scalar wArA = 1e20, wArAold;
        #pragma omp barrier
        #pragma omp master
        t0 = omp_get_wtime();
        for (int it = 0; it < nIter; it++) {
            wArAold = wArA;
            scalar acc = 0;
            for (label i = 0; i < m->nCells; i++) { const scalar w = rD[o + i] * rA[o + i]; wA[o + i] = w; acc += w * rA[o + i]; }
            part[d * 8] = acc;
            #pragma omp barrier
            wArA = 0; for (int k = 0; k < D; k++) wArA += part[k * 8];           /* reduce(sumOp) in rank order */
            if (it == 0) for (label i = 0; i < m->nCells; i++) pA[o + i] = wA[o + i];
            else { const scalar beta = wArA / wArAold; for (label i = 0; i < m->nCells; i++) pA[o + i] = wA[o + i] + beta * pA[o + i]; }
            #pragma omp barrier                                                   /* neighbours' pA must be complete */
            dom_amul_faceloop(s, d, pA, wA);
            acc = 0;
            for (label i = 0; i < m->nCells; i++) acc += wA[o + i] * pA[o + i];
            part[d * 8 + 1] = acc;
            #pragma omp barrier
            scalar wApA = 0; for (int k = 0; k < D; k++) wApA += part[k * 8 + 1];
            const scalar alpha = wArA / wApA;
            acc = 0;
            for (label i = 0; i < m->nCells; i++) {
                psi[o + i] += alpha * pA[o + i];
                const scalar r = rA[o + i] - alpha * wA[o + i];
                rA[o + i] = r; acc += fabs(r);
            }
            part[d * 8 + 2] = acc;
            #pragma omp barrier
        }
        #pragma omp master
        t1 = omp_get_wtime();
    }
    scalar res = 0; for (int k = 0; k < D; k++) res += part[k * 8 + 2];
    if (seconds) *seconds = t1 - t0;
    if (residual) *residual = res;
    free(psi); free(pA); free(wA); free(rA); free(rD); free(part);
    return nIter;
}
