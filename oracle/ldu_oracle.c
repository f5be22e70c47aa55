/*
 * ldu_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the lduMatrix hot path of SimFlowCFD/RapidCFD-dev,
 * used only as the parity checker by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py.  Nothing in the product path (the HIP engine
 * under rapidcfd-dev_amd/) may call into this file.
 *
 * PARITY UNPINNED: the reference tree holds no tests, tutorials, golden
 * vectors or benchmarks and cannot be compiled in
 * this container (needs nvcc + Thrust + a CUDA device).  The oracle is
 * therefore pinned by (i) algebraic identities checked in
 * tests/test_oracle.py (dense recomputation in long double, x'Ay == y'Ax,
 * Tmul(A) == Amul(A'), face-loop order vs row-gather order), (ii) the
 * golden fixtures it generated itself (tests/golden/golden_v1/v2.npz), and
 * (iii) THE REFERENCE'S OWN SOURCES wherever they are host code: PCG.C,
 * PBiCG.C, PBiCGStab.C, smoothSolver.C (+ their functor headers),
 * GAMGSolverSolve.C, GAMGSolverScale.C, the GAMG inter-level functor
 * headers, pairGAMGAgglomerate.C, and the SpMV family itself --
 * lduMatrixATmul.C and lduMatrixOperations.C with
 * lduAddressingFunctors.H, run on the host over a sequential stand-in for
 * Thrust -- are compiled from /root/reference against oracle/ref_shim/
 * into oracle/_ref/ (Makefile target `ref`); the solver loops here and the
 * V-cycle in gamg_oracle.c reproduce the reference's solve()/Vcycle()
 * functions bit for bit, sumA / H / H1 / sumDiag & co. and the literal
 * Amul reading give the bits of the reference's text
 * (tests/golden/golden_ref_*.npz, make_golden_ref.py).  Still an
 * assumption: which multiply-add pairs nvcc fuses inside the device row
 * functors (see orc_amul_functor_literal; the two readings differ by
 * < 2 ulp of the row magnitude).
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/src/OpenFOAM/matrices/lduMatrix/ unless noted).
 *
 * Conventions (reference defaults): scalar = IEEE double (etc/bashrc:76),
 * label = int32 (primitives/ints/label/label.H:57-66).
 *
 * A "system" is D >= 1 sub-domains solved in lock-step, which emulates the
 * reference's one-MPI-rank-per-GPU runs inside one process: processor
 * interfaces gather from the neighbour domain's psi, global sums add the
 * per-domain partial sums in rank order.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "ldu_oracle.h"


/* SolverPerformance.H:269-275 */
#define ORC_GREAT 1e20
#define ORC_SMALL 1e-20
#define ORC_VSMALL 1e-300 /* doubleScalarVSMALL */

/* ------------------------------------------------------------------------ */
/* data model                                                               */
/* ------------------------------------------------------------------------ */

/* ------------------------------------------------------------------------ */
/* lduAddressing derived tables                                             */
/* lduAddressing/lduAddressing.C:169-199 (losort = stable argsort of upper), */
/* :202-267 (ownerStart), :270-344 (losortStart)                            */
/* ------------------------------------------------------------------------ */
void orc_addr_tables(label nCells, label nFaces, const label *lower,
                     const label *upper, label *losort, label *ownerStart,
                     label *losortStart)
{
    label c, f;
    for (c = 0; c <= nCells; c++) { ownerStart[c] = 0; losortStart[c] = 0; }
    for (f = 0; f < nFaces; f++) { ownerStart[lower[f] + 1]++; losortStart[upper[f] + 1]++; }
    for (c = 0; c < nCells; c++) {
        ownerStart[c + 1] += ownerStart[c];
        losortStart[c + 1] += losortStart[c];
    }
    /* stable counting sort by upper[f] == thrust::stable_sort_by_key */
    label *cursor = (label *)malloc(sizeof(label) * (size_t)(nCells + 1));
    memcpy(cursor, losortStart, sizeof(label) * (size_t)(nCells + 1));
    for (f = 0; f < nFaces; f++) losort[cursor[upper[f]]++] = f;
    free(cursor);
}

/* ------------------------------------------------------------------------ */
/* system construction                                                      */
/* ------------------------------------------------------------------------ */
orc_system *orc_sys_create(int nDomains)
{
    orc_system *s = (orc_system *)calloc(1, sizeof(orc_system));
    s->nDomains = nDomains;
    s->dom = (orc_domain *)calloc((size_t)nDomains, sizeof(orc_domain));
    s->accurate_sums = 1;
    return s;
}

static void *dupmem(const void *p, size_t n)
{
    void *q = malloc(n ? n : 1);
    if (n) memcpy(q, p, n);
    return q;
}

/* lowerC may be NULL => symmetric (lduMatrix.C:328-345: lower aliases upper) */
void orc_sys_set_domain(orc_system *s, int d, label nCells, label nFaces,
                        const label *lower, const label *upper,
                        const scalar *diag, const scalar *lowerC,
                        const scalar *upperC)
{
    orc_domain *m = &s->dom[d];
    m->nCells = nCells; m->nFaces = nFaces;
    m->lower = (label *)dupmem(lower, sizeof(label) * (size_t)nFaces);
    m->upper = (label *)dupmem(upper, sizeof(label) * (size_t)nFaces);
    m->losort = (label *)malloc(sizeof(label) * (size_t)(nFaces ? nFaces : 1));
    m->ownerStart = (label *)malloc(sizeof(label) * (size_t)(nCells + 1));
    m->losortStart = (label *)malloc(sizeof(label) * (size_t)(nCells + 1));
    orc_addr_tables(nCells, nFaces, m->lower, m->upper, m->losort, m->ownerStart, m->losortStart);
    m->diag = (scalar *)dupmem(diag, sizeof(scalar) * (size_t)nCells);
    m->upperC = (scalar *)dupmem(upperC, sizeof(scalar) * (size_t)nFaces);
    m->symmetric = (lowerC == NULL);
    m->lowerC = m->symmetric ? m->upperC : (scalar *)dupmem(lowerC, sizeof(scalar) * (size_t)nFaces);
    int64_t off = 0; int i;
    for (i = 0; i < s->nDomains; i++) { s->dom[i].offset = off; off += s->dom[i].nCells; }
    s->nTotal = off;
}

/* replace coefficients of an existing domain (same addressing) */
void orc_sys_set_coeffs(orc_system *s, int d, const scalar *diag,
                        const scalar *lowerC, const scalar *upperC)
{
    orc_domain *m = &s->dom[d];
    memcpy(m->diag, diag, sizeof(scalar) * (size_t)m->nCells);
    memcpy(m->upperC, upperC, sizeof(scalar) * (size_t)m->nFaces);
    if (!m->symmetric && lowerC) memcpy(m->lowerC, lowerC, sizeof(scalar) * (size_t)m->nFaces);
}

int orc_sys_add_interface(orc_system *s, int d, label nbrDomain, label nbrPatch,
                          label nFaces, const label *faceCells,
                          const scalar *bouCoeffs, const scalar *intCoeffs)
{
    orc_domain *m = &s->dom[d];
    m->ifaces = (orc_iface *)realloc(m->ifaces, sizeof(orc_iface) * (size_t)(m->nIfaces + 1));
    orc_iface *p = &m->ifaces[m->nIfaces];
    p->nbrDomain = nbrDomain; p->nbrPatch = nbrPatch; p->nFaces = nFaces;
    p->faceCells = (label *)dupmem(faceCells, sizeof(label) * (size_t)nFaces);
    p->bouCoeffs = (scalar *)dupmem(bouCoeffs, sizeof(scalar) * (size_t)nFaces);
    p->intCoeffs = (scalar *)dupmem(intCoeffs, sizeof(scalar) * (size_t)nFaces);
    p->amiStart = p->amiAddr = NULL; p->amiW = NULL; p->amiLow = NULL; p->amiMagSf = NULL; p->factor = 1.0;
    p->nAmiParts = 0; p->amiPartDomain = p->amiPartPatch = p->amiPartStart = NULL;
    return m->nIfaces++;
}

/* make interface p of domain d a cyclicAMI one (weights into the faces of its neighbour interface); low may be NULL */
void orc_sys_set_iface_ami(orc_system *s, int d, int p, const label *start, const label *addr, const scalar *w, const unsigned char *low)
{
    orc_iface *q = &s->dom[d].ifaces[p];
    const label n = q->nFaces, na = start[n];
    q->amiStart = (label *)dupmem(start, sizeof(label) * (size_t)(n + 1));
    q->amiAddr = (label *)dupmem(addr, sizeof(label) * (size_t)(na ? na : 1));
    q->amiW = (scalar *)dupmem(w, sizeof(scalar) * (size_t)(na ? na : 1));
    q->amiLow = low ? (unsigned char *)dupmem(low, (size_t)(n ? n : 1)) : NULL;
}
/* the partner side of cyclicAMI interface p is split over nParts interfaces (every domain's interfaces must have been added): its
 * addresses number their faces concatenated in this order */
void orc_sys_set_iface_ami_parts(orc_system *s, int d, int p, int nParts, const label *partDomain, const label *partPatch)
{
    orc_iface *q = &s->dom[d].ifaces[p];
    int k;
    free(q->amiPartDomain); free(q->amiPartPatch); free(q->amiPartStart);
    q->nAmiParts = nParts;
    q->amiPartDomain = (label *)dupmem(partDomain, sizeof(label) * (size_t)nParts);
    q->amiPartPatch = (label *)dupmem(partPatch, sizeof(label) * (size_t)nParts);
    q->amiPartStart = (label *)malloc(sizeof(label) * (size_t)(nParts + 1));
    q->amiPartStart[0] = 0;
    for (k = 0; k < nParts; k++) q->amiPartStart[k + 1] = q->amiPartStart[k] + s->dom[partDomain[k]].ifaces[partPatch[k]].nFaces;
}
void orc_sys_set_iface_magsf(orc_system *s, int d, int p, const scalar *magSf)
{
    orc_iface *q = &s->dom[d].ifaces[p];
    free(q->amiMagSf);
    q->amiMagSf = (scalar *)dupmem(magSf, sizeof(scalar) * (size_t)(q->nFaces ? q->nFaces : 1));
}
/* transformCoupleField factor of interface p (cyclicLduInterfaceField.C:45-62, processorGAMGInterfaceField.C:213) */
void orc_sys_set_iface_transform(orc_system *s, int d, int p, scalar factor) { s->dom[d].ifaces[p].factor = factor; }

void orc_sys_set_accurate(orc_system *s, int on) { s->accurate_sums = on; }
int64_t orc_sys_size(const orc_system *s) { return s->nTotal; }

void orc_sys_destroy(orc_system *s)
{
    int d, i;
    for (d = 0; d < s->nDomains; d++) {
        orc_domain *m = &s->dom[d];
        free(m->lower); free(m->upper); free(m->losort); free(m->ownerStart);
        free(m->losortStart); free(m->diag); free(m->upperC);
        if (!m->symmetric) free(m->lowerC);
        for (i = 0; i < m->nIfaces; i++) {
            free(m->ifaces[i].faceCells); free(m->ifaces[i].bouCoeffs); free(m->ifaces[i].intCoeffs);
            free(m->ifaces[i].amiStart); free(m->ifaces[i].amiAddr); free(m->ifaces[i].amiW); free(m->ifaces[i].amiLow); free(m->ifaces[i].amiMagSf);
            free(m->ifaces[i].amiPartDomain); free(m->ifaces[i].amiPartPatch); free(m->ifaces[i].amiPartStart);
        }
        free(m->ifaces);
    }
    free(s->dom); free(s);
}

/* ------------------------------------------------------------------------ */
/* interfaces                                                               */
/* init: gather the neighbour's patch-internal field                        */
/*   (fvPatchTemplates.C:49-63 patchInternalField; processorFvPatchScalar-  */
/*    Field.C:36-118 send == the neighbour's gathered values arrive here)   */
/* update: result[faceCells[i]] -= coeffs[i]*pnf[i]  (negate=false)         */
/*   (lduAddressingFunctors.H:237-262 matrixInterfaceFunctor "yes the sign  */
/*    is correct"; coupledFvPatchField.C:236-257)                           */
/* negate=true adds instead (JacobiSmoother.C:75-93).                       */
/* ------------------------------------------------------------------------ */
static void update_interfaces(const orc_system *s, int d, int useIntCoeffs,
                              scalar coeffSign, const scalar *psiAll,
                              scalar *resultDom, int negate)
{
    const orc_domain *m = &s->dom[d];
    int p; label i;
    for (p = 0; p < m->nIfaces; p++) {
        const orc_iface *me = &m->ifaces[p];
        const orc_domain *nb = &s->dom[me->nbrDomain];
        const orc_iface *ot = &nb->ifaces[me->nbrPatch];
        const scalar *psiN = psiAll + nb->offset;
        const scalar *co = useIntCoeffs ? me->intCoeffs : me->bouCoeffs;
        for (i = 0; i < me->nFaces; i++) {
            /* one fused multiply-add per face, like every other term of the row
             * (the contraction nvcc applies to  result -= coeffs*pnf  in the functor) */
            const scalar cs = coeffSign * co[i];
            scalar pn;
            if (me->amiStart) {
                /* pnf = neighbour internal field, transformCoupleField (*= factor), then AMI interpolation:
                 * AMIInterpolationF.H:62-105 with multiplyWeightedOp<plusEqOp> (out += w*f, address order, contracted
                 * to one fma per term like the other device functors); low-weight faces take their own cell's value  */
                if (me->amiLow && me->amiLow[i]) pn = psiAll[m->offset + me->faceCells[i]];
                else {
                    label k; pn = 0.0;
                    for (k = me->amiStart[i]; k < me->amiStart[i + 1]; k++) {
                        int dq; label fq;
                        const orc_iface *pq = orc_ami_partner(s, me, me->amiAddr[k], &dq, &fq);   /* the one partner, or the split side's piece */
                        const scalar t = me->factor * psiAll[s->dom[dq].offset + pq->faceCells[fq]];
                        pn = fma(me->amiW[k], t, pn);
                    }
                }
            } else pn = me->factor * psiN[ot->faceCells[i]]; /* factor 1: exact */
            scalar *r = &resultDom[me->faceCells[i]];
            *r = negate ? fma(cs, pn, *r) : fma(-cs, pn, *r);
        }
    }
}


/* Speed only (the 216^3 parity tests run this oracle for >1000 iterations): rows / vector elements are independent, so
 * an OpenMP loop over them gives the same bits as the serial loop.  The long-double sums of domains above 2^20 cells are
 * evaluated in fixed blocks of 65536 elements (block sums in element order, then the block sums in block order, all in
 * long double): independent of the thread count, and equal to the plain left-to-right long-double sum to ~1e-19
 * relative; every golden fixture and small test case stays on the plain loop.                                        */
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_PAR_ROWS(n) ORC_PRAGMA(omp parallel for schedule(static) if ((n) > 200000))
#define ORC_PAR_ROWS_J(n) ORC_PRAGMA(omp parallel for schedule(static) private(j) if ((n) > 200000))
#define ORC_BLOCKED_SUM_MIN (1 << 20)
static long double blocked_ld_sum(const scalar *x, const scalar *y, int64_t n, int kind) /* 0: sum x, 1: sum x*y, 2: sum |x| */
{
    const int64_t B = 65536, nb = (n + B - 1) / B;
    long double *part = (long double *)malloc(sizeof(long double) * (size_t)nb), tot = 0;
    int64_t b;
    ORC_PRAGMA(omp parallel for schedule(static))
    for (b = 0; b < nb; b++) {
        const int64_t i0 = b * B, i1 = (i0 + B < n) ? i0 + B : n; int64_t i;
        long double acc = 0;
        if (kind == 0) for (i = i0; i < i1; i++) acc += x[i];
        else if (kind == 1) for (i = i0; i < i1; i++) acc += (long double)x[i] * (long double)y[i];
        else for (i = i0; i < i1; i++) acc += fabsl((long double)x[i]);
        part[b] = acc;
    }
    for (b = 0; b < nb; b++) tot += part[b];
    free(part);
    return tot;
}

/* ------------------------------------------------------------------------ */
/* Amul / Tmul: row-gather in the reference's summation order               */
/* lduMatrix/lduMatrixATmul.C:42-138 (matrixMultiplyFunctor<fast,3>):       */
/*   out = diag*psi; then upper faces of the row ascending; then lower      */
/*   faces in losort order.  Each term is folded with one fused             */
/*   multiply-add (the contraction nvcc applies by default); the HIP        */
/*   engine uses the same chain so both are bit-identical.                  */
/* Tmul (:264-342) swaps the roles of the lower and upper coefficients and  */
/* uses interfaceIntCoeffs.                                                 */
/* ------------------------------------------------------------------------ */
static void dom_mul_rows(const orc_domain *m, const scalar *Lower,
                         const scalar *Upper, const scalar *psi, scalar *Apsi)
{
    label c, j;
    ORC_PAR_ROWS_J(m->nCells)
    for (c = 0; c < m->nCells; c++) {
        scalar out = m->diag[c] * psi[c];
        for (j = m->ownerStart[c]; j < m->ownerStart[c + 1]; j++)
            out = fma(Upper[j], psi[m->upper[j]], out);
        for (j = m->losortStart[c]; j < m->losortStart[c + 1]; j++) {
            label f = m->losort[j];
            out = fma(Lower[f], psi[m->lower[f]], out);
        }
        Apsi[c] = out;
    }
}

void orc_amul(const orc_system *s, const scalar *psi, scalar *Apsi)
{
    int d;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        dom_mul_rows(m, m->lowerC, m->upperC, psi + m->offset, Apsi + m->offset);
        update_interfaces(s, d, 0, 1.0, psi, Apsi + m->offset, 0);
    }
}

void orc_tmul(const orc_system *s, const scalar *psi, scalar *Tpsi)
{
    int d;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        dom_mul_rows(m, m->upperC, m->lowerC, psi + m->offset, Tpsi + m->offset);
        update_interfaces(s, d, 1, 1.0, psi, Tpsi + m->offset, 0);
    }
}

/* The reference functor read LITERALLY (lduMatrixATmul.C:42-138, matrixMultiplyFunctor<fast,3>): the products of the first
 * three own-side and the first three neighbour-side faces are staged in tmpSum[] behind `if (i < size)` guards and then
 * added to out = diag*psi one by one; further own-side faces are folded as out += upper*psi; further neighbour-side faces
 * go to a separate accumulator nExtra that is added last.  Which of these adds nvcc fuses with their multiplies cannot be
 * established without the CUDA binary: a product that reaches its add through a guarded array element (the staged six)
 * most likely stays separately rounded, the `+= a*b` statements of the extras most likely become fma.  orc_amul above uses
 * ONE fma per term in row order (own faces ascending, then neighbour faces in losort order) -- the same order for rows
 * with at most 3+3 faces (every hex mesh), a different association for longer rows.  This literal variant exists so the
 * difference is measured instead of assumed (tests/test_oracle.py::test_functor_literal_variant_is_within_rounding).   */
void orc_amul_functor_literal(const orc_system *s, const scalar *psiAll, scalar *ApsiAll)
{
    int d;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *psi = psiAll + m->offset; scalar *Apsi = ApsiAll + m->offset;
        label c, i;
        for (c = 0; c < m->nCells; c++) {
            const label oStart = m->ownerStart[c], oSize = m->ownerStart[c + 1] - oStart;
            const label nStart = m->losortStart[c], nSize = m->losortStart[c + 1] - nStart;
            scalar tmpSum[6] = {0, 0, 0, 0, 0, 0}, nExtra = 0;
            volatile scalar prod;                      /* keeps the staged products separately rounded under any compiler flags */
            scalar out = m->diag[c] * psi[c];
            for (i = 0; i < 3; i++) if (i < oSize) { prod = m->upperC[oStart + i] * psi[m->upper[oStart + i]]; tmpSum[i] = prod; }
            for (i = 0; i < 3; i++) if (i < nSize) { const label f = m->losort[nStart + i]; prod = m->lowerC[f] * psi[m->lower[f]]; tmpSum[i + 3] = prod; }
            for (i = 0; i < 6; i++) out += tmpSum[i];
            for (i = 3; i < oSize; i++) out = fma(m->upperC[oStart + i], psi[m->upper[oStart + i]], out);
            for (i = 3; i < nSize; i++) { const label f = m->losort[nStart + i]; nExtra = fma(m->lowerC[f], psi[m->lower[f]], nExtra); }
            Apsi[c] = out + nExtra;
        }
        update_interfaces(s, d, 0, 1.0, psiAll, Apsi, 0);
    }
}

/* Upstream OpenFOAM CPU face loop (OpenFOAM-2.3.x lduMatrixATmul.C, not in the
 * reference tree):
 *   Apsi[c]=D[c]psi[c]; for f: Apsi[u[f]]+=L[f]psi[l[f]]; Apsi[l[f]]+=U[f]psi[u[f]]
 * Used for the cpu_baseline and as an order-independent cross-check.        */
void orc_amul_faceloop(const orc_system *s, const scalar *psi, scalar *Apsi)
{
    int d; label c, f;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *x = psi + m->offset; scalar *y = Apsi + m->offset;
        for (c = 0; c < m->nCells; c++) y[c] = m->diag[c] * x[c];
        for (f = 0; f < m->nFaces; f++) {
            y[m->upper[f]] += m->lowerC[f] * x[m->lower[f]];
            y[m->lower[f]] += m->upperC[f] * x[m->upper[f]];
        }
        update_interfaces(s, d, 0, 1.0, psi, y, 0);
    }
}

/* sumA: lduMatrixATmul.C:345-395 -- diag + sum(upper of own faces) +
 * sum(lower of neighbour faces), then minus interface bouCoeffs.           */
void orc_sumA(const orc_system *s, scalar *sumA)
{
    int d, p; label c, j, i;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        scalar *o = sumA + m->offset;
        ORC_PAR_ROWS_J(m->nCells)
        for (c = 0; c < m->nCells; c++) {
            scalar out = m->diag[c];
            for (j = m->ownerStart[c]; j < m->ownerStart[c + 1]; j++) out += m->upperC[j];
            for (j = m->losortStart[c]; j < m->losortStart[c + 1]; j++) out += m->lowerC[m->losort[j]];
            o[c] = out;
        }
        for (p = 0; p < m->nIfaces; p++)
            for (i = 0; i < m->ifaces[p].nFaces; i++)
                o[m->ifaces[p].faceCells[i]] += -m->ifaces[p].bouCoeffs[i];
    }
}

/* residual: lduMatrixATmul.C:397-496 -- rA = source - diag*psi - sum(upper*psi[u])
 * - sum(lower*psi[l]), interfaces with negated bouCoeffs (mBouCoeffs).      */
void orc_residual(const orc_system *s, const scalar *psi, const scalar *source, scalar *rA)
{
    int d; label c, j;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *x = psi + m->offset; scalar *r = rA + m->offset;
        ORC_PAR_ROWS_J(m->nCells)
        for (c = 0; c < m->nCells; c++) {
            scalar out = source[m->offset + c] - m->diag[c] * x[c];
            for (j = m->ownerStart[c]; j < m->ownerStart[c + 1]; j++)
                out = fma(-m->upperC[j], x[m->upper[j]], out);
            for (j = m->losortStart[c]; j < m->losortStart[c + 1]; j++) {
                label f = m->losort[j];
                out = fma(-m->lowerC[f], x[m->lower[f]], out);
            }
            r[c] = out;
        }
        update_interfaces(s, d, 0, -1.0, psi, r, 0);
    }
}

/* H1 / H / faceH: lduMatrixATmul.C:533-554, lduMatrixOperations.C:130-154
 * (lduMatrixTemplates.C:52-86), lduMatrixTemplates.C:110-148               */
void orc_H(const orc_system *s, const scalar *psi, scalar *H)
{
    int d; label c, j;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *x = psi + m->offset; scalar *h = H + m->offset;
        for (c = 0; c < m->nCells; c++) {
            scalar out = 0;
            for (j = m->ownerStart[c]; j < m->ownerStart[c + 1]; j++)
                out = fma(-m->upperC[j], x[m->upper[j]], out);
            for (j = m->losortStart[c]; j < m->losortStart[c + 1]; j++) {
                label f = m->losort[j];
                out = fma(-m->lowerC[f], x[m->lower[f]], out);
            }
            h[c] = out;
        }
    }
}

void orc_H1(const orc_system *s, scalar *H1)
{
    int d; label c, j;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        scalar *h = H1 + m->offset;
        for (c = 0; c < m->nCells; c++) {
            scalar out = 0;
            for (j = m->ownerStart[c]; j < m->ownerStart[c + 1]; j++) out -= m->upperC[j];
            for (j = m->losortStart[c]; j < m->losortStart[c + 1]; j++) out -= m->lowerC[m->losort[j]];
            h[c] = out;
        }
    }
}

/* faceH[f] = upper[f]*psi[u[f]] - lower[f]*psi[l[f]]  (single domain d)    */
void orc_faceH(const orc_system *s, int d, const scalar *psi, scalar *faceH)
{
    const orc_domain *m = &s->dom[d]; label f;
    const scalar *x = psi + m->offset;
    for (f = 0; f < m->nFaces; f++)
        faceH[f] = fma(m->upperC[f], x[m->upper[f]], -(m->lowerC[f] * x[m->lower[f]]));   /* lduMatrixfaceHFunctor as compiled (oracle/_ref/libref_fvm.so) */
}

/* sumDiag / negSumDiag / sumMagOffDiag: lduMatrixOperations.C:36-106       */
void orc_neg_sum_diag(label nCells, label nFaces, const label *lower, const label *upper,
                      const scalar *lowerC, const scalar *upperC, scalar *diag)
{
    /* row order of the reference (K2): own faces (subtract upper), then
     * neighbour faces in losort order (subtract lower)                     */
    label *losort = (label *)malloc(sizeof(label) * (size_t)(nFaces ? nFaces : 1));
    label *os = (label *)malloc(sizeof(label) * (size_t)(nCells + 1));
    label *ls = (label *)malloc(sizeof(label) * (size_t)(nCells + 1));
    label c, j;
    orc_addr_tables(nCells, nFaces, lower, upper, losort, os, ls);
    for (c = 0; c < nCells; c++) {
        scalar out = diag[c];
        /* negSumDiag: Diag[l[face]] -= Lower[face]; Diag[u[face]] -= Upper[face]
         * (lduMatrixOperations.C:62-83: owner rows take lower, neighbour rows upper) */
        for (j = os[c]; j < os[c + 1]; j++) out -= lowerC[j];
        for (j = ls[c]; j < ls[c + 1]; j++) out -= upperC[losort[j]];
        diag[c] = out;
    }
    free(losort); free(os); free(ls);
}

/* ------------------------------------------------------------------------ */
/* reductions (gpuFieldCommonFunctions.C:351-367,420-440,492-511,585-634):  */
/* per-domain local sums, then summed over domains in rank order            */
/* (MPI_Allreduce, allReduceTemplates.C:195-208).  The reference's          */
/* thrust::reduce tree is unspecified; the oracle evaluates the local sums  */
/* in long double so that it sits at the correctly-rounded value any tree   */
/* converges to (accurate_sums=1), or in plain double for timing runs.      */
/* ------------------------------------------------------------------------ */
static scalar g_sum_prod(const orc_system *s, const scalar *a, const scalar *b)
{
    scalar total = 0; int d; label i;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *x = a + m->offset, *y = b + m->offset;
        if (s->accurate_sums) {
            long double acc = 0;
            if (m->nCells > ORC_BLOCKED_SUM_MIN) acc = blocked_ld_sum(x, y, m->nCells, 1);
            else for (i = 0; i < m->nCells; i++) acc += (long double)x[i] * (long double)y[i];
            total += (scalar)acc;
        } else {
            scalar a0 = 0, a1 = 0, a2 = 0, a3 = 0; label n4 = m->nCells & ~3;
            for (i = 0; i < n4; i += 4) { a0 += x[i]*y[i]; a1 += x[i+1]*y[i+1]; a2 += x[i+2]*y[i+2]; a3 += x[i+3]*y[i+3]; }
            for (; i < m->nCells; i++) a0 += x[i] * y[i];
            total += (a0 + a1) + (a2 + a3);
        }
    }
    return total;
}

static scalar g_sum_mag(const orc_system *s, const scalar *a)
{
    scalar total = 0; int d; label i;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *x = a + m->offset;
        if (s->accurate_sums) {
            long double acc = 0;
            if (m->nCells > ORC_BLOCKED_SUM_MIN) acc = blocked_ld_sum(x, NULL, m->nCells, 2);
            else for (i = 0; i < m->nCells; i++) acc += fabsl((long double)x[i]);
            total += (scalar)acc;
        } else {
            scalar a0 = 0, a1 = 0, a2 = 0, a3 = 0; label n4 = m->nCells & ~3;
            for (i = 0; i < n4; i += 4) { a0 += fabs(x[i]); a1 += fabs(x[i+1]); a2 += fabs(x[i+2]); a3 += fabs(x[i+3]); }
            for (; i < m->nCells; i++) a0 += fabs(x[i]);
            total += (a0 + a1) + (a2 + a3);
        }
    }
    return total;
}

static scalar g_sum(const orc_system *s, const scalar *a)
{
    scalar total = 0; int d; label i;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *x = a + m->offset;
        long double acc = 0;
        if (m->nCells > ORC_BLOCKED_SUM_MIN) acc = blocked_ld_sum(x, NULL, m->nCells, 0);
        else for (i = 0; i < m->nCells; i++) acc += x[i];
        total += (scalar)acc;
    }
    return total;
}

scalar orc_gSumProd(const orc_system *s, const scalar *a, const scalar *b) { return g_sum_prod(s, a, b); }
scalar orc_gSumMag(const orc_system *s, const scalar *a) { return g_sum_mag(s, a); }
scalar orc_gSum(const orc_system *s, const scalar *a) { return g_sum(s, a); }

/* normFactor: lduMatrix/lduMatrixSolver.C:182-236
 *   sumA; xRef = gAverage(psi); sum(|Apsi - xRef*sumA| + |source - xRef*sumA|) + small */
scalar orc_norm_factor(const orc_system *s, const scalar *psi, const scalar *source,
                       const scalar *Apsi, scalar *tmp)
{
    orc_sumA(s, tmp);
    scalar average = g_sum(s, psi) / (scalar)s->nTotal; /* gAverage :611-634 */
    scalar total = 0; int d; label i;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        long double acc = 0;
        for (i = 0; i < m->nCells; i++) {
            int64_t k = m->offset + i;
            scalar tmpVal = average * tmp[k];
            acc += (long double)(fabs(Apsi[k] - tmpVal) + fabs(source[k] - tmpVal));
        }
        total += (scalar)acc;
    }
    return total + ORC_SMALL;
}

/* ------------------------------------------------------------------------ */
/* preconditioners                                                          */
/* ------------------------------------------------------------------------ */
enum { ORC_PRECOND_NONE = 0, ORC_PRECOND_DIAGONAL = 1, ORC_PRECOND_AINV = 2,
       ORC_PRECOND_DIC_UPSTREAM = 3, ORC_PRECOND_DILU_UPSTREAM = 4 };

typedef struct {
    int kind;
    scalar *rD; /* [nTotal] */
} orc_precond;

/* diagonalPreconditioner.C:45-68 (rD = 1/diag);  AINVPreconditioner.C:18-42 same.
 * DIC/DILU upstream (OpenFOAM-2.3.x DICPreconditioner.C / DILUPreconditioner.C
 * calcReciprocalD; not in the reference tree).        */
static orc_precond *precond_new(const orc_system *s, int kind)
{
    orc_precond *P = (orc_precond *)calloc(1, sizeof(orc_precond));
    P->kind = kind;
    P->rD = (scalar *)malloc(sizeof(scalar) * (size_t)(s->nTotal ? s->nTotal : 1));
    int d; label i, f;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        scalar *rD = P->rD + m->offset;
        if (kind == ORC_PRECOND_DIC_UPSTREAM || kind == ORC_PRECOND_DILU_UPSTREAM) {
            for (i = 0; i < m->nCells; i++) rD[i] = m->diag[i];
            for (f = 0; f < m->nFaces; f++)
                rD[m->upper[f]] -= m->upperC[f] * m->lowerC[f] / rD[m->lower[f]];
            for (i = 0; i < m->nCells; i++) rD[i] = 1.0 / rD[i];
        } else {
            for (i = 0; i < m->nCells; i++) rD[i] = 1.0 / m->diag[i];
        }
    }
    return P;
}

static void precond_free(orc_precond *P) { free(P->rD); free(P); }

/* AINV: AINVPreconditionerF.H:41-99
 *   w[c] = rD[c]*(r[c] - sum_own upper*rD[n]*r[n] - sum_nei lower*rD[o]*r[o])
 * transpose variant swaps lower/upper (AINVPreconditioner.C:64-70).        */
static void ainv_apply(const orc_domain *m, const scalar *rD, const scalar *Lower,
                       const scalar *Upper, const scalar *r, scalar *w)
{
    label c, j;
    ORC_PAR_ROWS_J(m->nCells)
    for (c = 0; c < m->nCells; c++) {
        scalar out = 0;
        for (j = m->ownerStart[c]; j < m->ownerStart[c + 1]; j++) {
            label n = m->upper[j];
            out = fma(Upper[j] * rD[n], r[n], out);
        }
        for (j = m->losortStart[c]; j < m->losortStart[c + 1]; j++) {
            label f = m->losort[j]; label o = m->lower[f];
            out = fma(Lower[f] * rD[o], r[o], out);
        }
        w[c] = rD[c] * (r[c] - out);
    }
}

static void precondition(const orc_system *s, const orc_precond *P, int transpose,
                         const scalar *r, scalar *w)
{
    int d; label i, f;
    for (d = 0; d < s->nDomains; d++) {
        const orc_domain *m = &s->dom[d];
        const scalar *rD = P->rD + m->offset; const scalar *rr = r + m->offset;
        scalar *ww = w + m->offset;
        switch (P->kind) {
        case ORC_PRECOND_NONE: /* noPreconditioner.C:66-71 */
            ORC_PAR_ROWS(m->nCells)
            for (i = 0; i < m->nCells; i++) ww[i] = rr[i];
            break;
        case ORC_PRECOND_DIAGONAL: /* diagonalPreconditioner.C:74-89 */
            ORC_PAR_ROWS(m->nCells)
            for (i = 0; i < m->nCells; i++) ww[i] = rD[i] * rr[i];
            break;
        case ORC_PRECOND_AINV:
            if (!transpose) ainv_apply(m, rD, m->lowerC, m->upperC, rr, ww);
            else            ainv_apply(m, rD, m->upperC, m->lowerC, rr, ww);
            break;
        case ORC_PRECOND_DIC_UPSTREAM: /* upstream DICPreconditioner::precondition */
            for (i = 0; i < m->nCells; i++) ww[i] = rD[i] * rr[i];
            for (f = 0; f < m->nFaces; f++)
                ww[m->upper[f]] -= rD[m->upper[f]] * m->upperC[f] * ww[m->lower[f]];
            for (f = m->nFaces - 1; f >= 0; f--)
                ww[m->lower[f]] -= rD[m->lower[f]] * m->upperC[f] * ww[m->upper[f]];
            break;
        case ORC_PRECOND_DILU_UPSTREAM: { /* upstream DILUPreconditioner */
            const scalar *Lo = transpose ? m->upperC : m->lowerC;
            const scalar *Up = transpose ? m->lowerC : m->upperC;
            for (i = 0; i < m->nCells; i++) ww[i] = rD[i] * rr[i];
            if (!transpose) {
                for (f = 0; f < m->nFaces; f++) {
                    label sf = m->losort[f];
                    ww[m->upper[sf]] -= rD[m->upper[sf]] * Lo[sf] * ww[m->lower[sf]];
                }
                for (f = m->nFaces - 1; f >= 0; f--)
                    ww[m->lower[f]] -= rD[m->lower[f]] * Up[f] * ww[m->upper[f]];
            } else {
                for (f = 0; f < m->nFaces; f++)
                    ww[m->upper[f]] -= rD[m->upper[f]] * Lo[f] * ww[m->lower[f]];
                for (f = m->nFaces - 1; f >= 0; f--) {
                    label sf = m->losort[f];
                    ww[m->lower[sf]] -= rD[m->lower[sf]] * Up[sf] * ww[m->upper[sf]];
                }
            }
        } break;
        }
    }
}

/* public: build + apply once (for unit parity tests of the HIP kernels) */
void orc_precondition(const orc_system *s, int kind, int transpose, const scalar *r, scalar *w)
{
    orc_precond *P = precond_new(s, kind);
    precondition(s, P, transpose, r, w);
    precond_free(P);
}

/* ------------------------------------------------------------------------ */
/* smoothers                                                                */
/* Jacobi (== the reference's "GaussSeidel"): JacobiSmoother.C:39-148,      */
/* JacobiSmootherF.H:50-108, omega default 0.9 (JacobiSmoother.C:34-36).    */
/*   per sweep: bPrime = source (+ interface contributions, negate=true);   */
/*   psi' = (1-w)psi + w*rD*b' - w*rD*sum(offdiag*psi[nbr]);  psi = psi'    */
/* ------------------------------------------------------------------------ */
void orc_jacobi_smooth(const orc_system *s, scalar omega, scalar *psi,
                       const scalar *source, int nSweeps)
{
    scalar *bP = (scalar *)malloc(sizeof(scalar) * (size_t)(s->nTotal ? s->nTotal : 1));
    scalar *nw = (scalar *)malloc(sizeof(scalar) * (size_t)(s->nTotal ? s->nTotal : 1));
    int sweep, d; label c, j;
    for (sweep = 0; sweep < nSweeps; sweep++) {
        memcpy(bP, source, sizeof(scalar) * (size_t)s->nTotal);
        for (d = 0; d < s->nDomains; d++)
            update_interfaces(s, d, 0, 1.0, psi, bP + s->dom[d].offset, 1);
        for (d = 0; d < s->nDomains; d++) {
            const orc_domain *m = &s->dom[d];
            const scalar *x = psi + m->offset; const scalar *b = bP + m->offset;
            scalar *y = nw + m->offset;
            for (c = 0; c < m->nCells; c++) {
                const scalar rD = 1.0 / m->diag[c];
                scalar extra = (1 - omega) * x[c] + omega * rD * b[c];
                scalar out = 0;
                for (j = m->ownerStart[c]; j < m->ownerStart[c + 1]; j++)
                    out = fma(m->upperC[j], x[m->upper[j]], out);
                for (j = m->losortStart[c]; j < m->losortStart[c + 1]; j++) {
                    label f = m->losort[j];
                    out = fma(m->lowerC[f], x[m->lower[f]], out);
                }
                y[c] = extra - omega * rD * out;
            }
        }
        memcpy(psi, nw, sizeof(scalar) * (size_t)s->nTotal);
    }
    free(bP); free(nw);
}

/* Upstream Gauss-Seidel (OpenFOAM-2.3.x GaussSeidelSmoother.C; not in the
 * reference tree).  Single-domain / interface-free restatement, used only for
 * the config-1 CPU plumbing case.                                          */
void orc_gauss_seidel_upstream(const orc_system *s, scalar *psi, const scalar *source, int nSweeps)
{
    int sweep, d; label c, f;
    for (sweep = 0; sweep < nSweeps; sweep++) {
        for (d = 0; d < s->nDomains; d++) {
            const orc_domain *m = &s->dom[d];
            scalar *x = psi + m->offset;
            scalar *bP = (scalar *)malloc(sizeof(scalar) * (size_t)m->nCells);
            memcpy(bP, source + m->offset, sizeof(scalar) * (size_t)m->nCells);
            update_interfaces(s, d, 0, 1.0, psi, bP, 1);
            for (c = 0; c < m->nCells; c++) {
                scalar cur = bP[c];
                for (f = m->ownerStart[c]; f < m->ownerStart[c + 1]; f++) cur -= m->upperC[f] * x[m->upper[f]];
                cur /= m->diag[c];
                for (f = m->ownerStart[c]; f < m->ownerStart[c + 1]; f++) bP[m->upper[f]] -= m->lowerC[f] * cur;
                x[c] = cur;
            }
            free(bP);
        }
    }
}

/* ------------------------------------------------------------------------ */
/* SolverPerformance (LduMatrix/LduMatrix/SolverPerformance.C:31-92)        */
/* ------------------------------------------------------------------------ */
typedef struct {
    scalar initialResidual, finalResidual, normFactor;
    int32_t nIterations, converged, singular;
} orc_perf;

typedef struct {
    scalar tolerance, relTol; /* lduMatrixSolver.C:167-173 defaults 1e-6 / 0 */
    int32_t maxIter, minIter; /* 1000 / 0 */
} orc_controls;

static int check_convergence(orc_perf *p, const orc_controls *c)
{
    p->converged = (p->finalResidual < c->tolerance) ||
                   (c->relTol > ORC_SMALL && p->finalResidual < c->relTol * p->initialResidual);
    return p->converged;
}

static int check_singularity(orc_perf *p, scalar v)
{
    p->singular = (v < ORC_VSMALL);
    return p->singular;
}

static scalar *valloc(const orc_system *s) { return (scalar *)calloc((size_t)(s->nTotal ? s->nTotal : 1), sizeof(scalar)); }

static void hist_put(scalar *hist, int histLen, int k, scalar v) { if (hist && k < histLen) hist[k] = v; }

/* Vector updates (lduMatrixSolverFunctors.H:7-44, lduMatrixFunctors.H:186-301) are
 * written as one fused multiply-add each: a*x+y contracts to an FMA under nvcc's
 * default -fmad=true, and the HIP engine issues the same fma explicitly.        */

/* PCG: solvers/PCG/PCG.C:68-208.  hist[k] = normalised residual after k
 * iterations (hist[0] = initial).                                           */
void orc_pcg_solve(const orc_system *s, scalar *psi, const scalar *source,
                   const orc_controls *ctl, int precondKind, orc_perf *perf,
                   scalar *hist, int histLen)
{
    int64_t n = s->nTotal, i;
    scalar *pA = valloc(s), *wA = valloc(s), *rA = valloc(s);
    scalar wArA = ORC_GREAT, wArAold = wArA;
    memset(perf, 0, sizeof(*perf));

    orc_amul(s, psi, wA);
    ORC_PAR_ROWS(n)
    for (i = 0; i < n; i++) rA[i] = source[i] - wA[i];
    scalar normFactor = orc_norm_factor(s, psi, source, wA, pA);
    perf->normFactor = normFactor;
    perf->initialResidual = g_sum_mag(s, rA) / normFactor;
    perf->finalResidual = perf->initialResidual;
    hist_put(hist, histLen, 0, perf->initialResidual);

    if (ctl->minIter > 0 || !check_convergence(perf, ctl)) {
        orc_precond *P = precond_new(s, precondKind);
        do {
            wArAold = wArA;
            precondition(s, P, 0, rA, wA);
            wArA = g_sum_prod(s, wA, rA);
            if (perf->nIterations == 0) {
                memcpy(pA, wA, sizeof(scalar) * (size_t)n);
            } else {
                scalar beta = wArA / wArAold;
                ORC_PAR_ROWS(n)
                for (i = 0; i < n; i++) pA[i] = fma(beta, pA[i], wA[i]);
            }
            orc_amul(s, pA, wA);
            scalar wApA = g_sum_prod(s, wA, pA);
            if (check_singularity(perf, fabs(wApA) / normFactor)) break;
            scalar alpha = wArA / wApA;
            ORC_PAR_ROWS(n)
            for (i = 0; i < n; i++) { psi[i] = fma(alpha, pA[i], psi[i]); rA[i] = fma(-alpha, wA[i], rA[i]); }
            perf->finalResidual = g_sum_mag(s, rA) / normFactor;
            hist_put(hist, histLen, perf->nIterations + 1, perf->finalResidual);
        } while ((perf->nIterations++ < ctl->maxIter && !check_convergence(perf, ctl)) ||
                 perf->nIterations < ctl->minIter);
        precond_free(P);
    }
    free(pA); free(wA); free(rA);
}

/* PBiCG: solvers/PBiCG/PBiCG.C:67-246 */
void orc_pbicg_solve(const orc_system *s, scalar *psi, const scalar *source,
                     const orc_controls *ctl, int precondKind, orc_perf *perf,
                     scalar *hist, int histLen)
{
    int64_t n = s->nTotal, i;
    scalar *pA = valloc(s), *pT = valloc(s), *wA = valloc(s), *wT = valloc(s), *rA = valloc(s), *rT = valloc(s);
    scalar wArT = ORC_GREAT, wArTold = wArT;
    memset(perf, 0, sizeof(*perf));

    orc_amul(s, psi, wA);
    orc_tmul(s, psi, wT);
    for (i = 0; i < n; i++) rA[i] = source[i] - wA[i];
    for (i = 0; i < n; i++) rT[i] = source[i] - wT[i];
    scalar normFactor = orc_norm_factor(s, psi, source, wA, pA);
    perf->normFactor = normFactor;
    perf->initialResidual = g_sum_mag(s, rA) / normFactor;
    perf->finalResidual = perf->initialResidual;
    hist_put(hist, histLen, 0, perf->initialResidual);

    if (ctl->minIter > 0 || !check_convergence(perf, ctl)) {
        orc_precond *P = precond_new(s, precondKind);
        do {
            wArTold = wArT;
            precondition(s, P, 0, rA, wA);
            precondition(s, P, 1, rT, wT);
            wArT = g_sum_prod(s, wA, rT);
            if (perf->nIterations == 0) {
                memcpy(pA, wA, sizeof(scalar) * (size_t)n);
                memcpy(pT, wT, sizeof(scalar) * (size_t)n);
            } else {
                scalar beta = wArT / wArTold;
                for (i = 0; i < n; i++) pA[i] = fma(beta, pA[i], wA[i]);
                for (i = 0; i < n; i++) pT[i] = fma(beta, pT[i], wT[i]);
            }
            orc_amul(s, pA, wA);
            orc_tmul(s, pT, wT);
            scalar wApT = g_sum_prod(s, wA, pT);
            if (check_singularity(perf, fabs(wApT) / normFactor)) break;
            scalar alpha = wArT / wApT;
            for (i = 0; i < n; i++) psi[i] = fma(alpha, pA[i], psi[i]);
            for (i = 0; i < n; i++) rA[i] = fma(-alpha, wA[i], rA[i]);
            for (i = 0; i < n; i++) rT[i] = fma(-alpha, wT[i], rT[i]);
            perf->finalResidual = g_sum_mag(s, rA) / normFactor;
            hist_put(hist, histLen, perf->nIterations + 1, perf->finalResidual);
        } while ((perf->nIterations++ < ctl->maxIter && !check_convergence(perf, ctl)) ||
                 perf->nIterations < ctl->minIter);
        precond_free(P);
    }
    free(pA); free(pT); free(wA); free(wT); free(rA); free(rT);
}

/* PBiCGStab: solvers/PBiCGStab/PBiCGStab.C:67-300.  replicateQuirk=1 keeps the
 * reference's `psi += omega*yA` (PBiCGStab.C:263-270);
 * 0 uses the textbook `psi += omega*zA`.                                    */
void orc_pbicgstab_solve(const orc_system *s, scalar *psi, const scalar *source,
                         const orc_controls *ctl, int precondKind, int replicateQuirk,
                         orc_perf *perf, scalar *hist, int histLen)
{
    int64_t n = s->nTotal, i;
    scalar *pA = valloc(s), *yA = valloc(s), *rA = valloc(s);
    memset(perf, 0, sizeof(*perf));

    orc_amul(s, psi, yA);
    for (i = 0; i < n; i++) rA[i] = source[i] - yA[i];
    scalar normFactor = orc_norm_factor(s, psi, source, yA, pA);
    perf->normFactor = normFactor;
    perf->initialResidual = g_sum_mag(s, rA) / normFactor;
    perf->finalResidual = perf->initialResidual;
    hist_put(hist, histLen, 0, perf->initialResidual);

    if (ctl->minIter > 0 || !check_convergence(perf, ctl)) {
        scalar *AyA = valloc(s), *sA = valloc(s), *zA = valloc(s), *tA = valloc(s), *rA0 = valloc(s);
        memcpy(rA0, rA, sizeof(scalar) * (size_t)n);
        scalar rA0rA = 0, alpha = 0, omega = 0;
        orc_precond *P = precond_new(s, precondKind);
        int early = 0;
        do {
            const scalar rA0rAold = rA0rA;
            rA0rA = g_sum_prod(s, rA0, rA);
            if (check_singularity(perf, fabs(rA0rA))) break;
            if (perf->nIterations == 0) {
                memcpy(pA, rA, sizeof(scalar) * (size_t)n);
            } else {
                if (check_singularity(perf, fabs(omega))) break;
                const scalar beta = (rA0rA / rA0rAold) * (alpha / omega);
                for (i = 0; i < n; i++) {
                    scalar result1 = fma(-omega, AyA[i], pA[i]);
                    pA[i] = fma(beta, result1, rA[i]);
                }
            }
            precondition(s, P, 0, pA, yA);
            orc_amul(s, yA, AyA);
            const scalar rA0AyA = g_sum_prod(s, rA0, AyA);
            alpha = rA0rA / rA0AyA;
            for (i = 0; i < n; i++) sA[i] = fma(-alpha, AyA[i], rA[i]);
            perf->finalResidual = g_sum_mag(s, sA) / normFactor;
            if (check_convergence(perf, ctl)) {
                for (i = 0; i < n; i++) psi[i] = fma(alpha, yA[i], psi[i]);
                perf->nIterations++;
                hist_put(hist, histLen, perf->nIterations, perf->finalResidual);
                early = 1;
                break;
            }
            precondition(s, P, 0, sA, zA);
            orc_amul(s, zA, tA);
            const scalar tAtA = g_sum_prod(s, tA, tA);
            omega = g_sum_prod(s, tA, sA) / tAtA;
            for (i = 0; i < n; i++) psi[i] = fma(alpha, yA[i], psi[i]);
            if (replicateQuirk) for (i = 0; i < n; i++) psi[i] = fma(omega, yA[i], psi[i]);
            else                for (i = 0; i < n; i++) psi[i] = fma(omega, zA[i], psi[i]);
            for (i = 0; i < n; i++) rA[i] = fma(-omega, tA[i], sA[i]);
            perf->finalResidual = g_sum_mag(s, rA) / normFactor;
            hist_put(hist, histLen, perf->nIterations + 1, perf->finalResidual);
        } while ((perf->nIterations++ < ctl->maxIter && !check_convergence(perf, ctl)) ||
                 perf->nIterations < ctl->minIter);
        (void)early;
        precond_free(P);
        free(AyA); free(sA); free(zA); free(tA); free(rA0);
    }
    free(pA); free(yA); free(rA);
}

/* smoothSolver: solvers/smoothSolver/smoothSolver.C -- nSweeps smoothing sweeps
 * between residual evaluations.                                             */
void orc_smooth_solve(const orc_system *s, scalar *psi, const scalar *source,
                      const orc_controls *ctl, scalar omega, int nSweeps,
                      orc_perf *perf, scalar *hist, int histLen)
{
    int64_t n = s->nTotal;
    scalar *Apsi = valloc(s), *tmp = valloc(s), *res = valloc(s);
    memset(perf, 0, sizeof(*perf));
    if (nSweeps < 0) { /* smoothSolver.C:87-110: negative nSweeps => fixed number of sweeps */
        orc_jacobi_smooth(s, omega, psi, source, -nSweeps);
        perf->nIterations -= nSweeps;
    } else {
        orc_amul(s, psi, Apsi);
        scalar normFactor = orc_norm_factor(s, psi, source, Apsi, tmp);
        perf->normFactor = normFactor;
        int64_t i;
        for (i = 0; i < n; i++) res[i] = source[i] - Apsi[i];
        perf->initialResidual = g_sum_mag(s, res) / normFactor;
        perf->finalResidual = perf->initialResidual;
        hist_put(hist, histLen, 0, perf->initialResidual);
        if (ctl->minIter > 0 || !check_convergence(perf, ctl)) {
            do {
                orc_jacobi_smooth(s, omega, psi, source, nSweeps);
                orc_residual(s, psi, source, res);
                perf->finalResidual = g_sum_mag(s, res) / normFactor;
                hist_put(hist, histLen, perf->nIterations / nSweeps + 1, perf->finalResidual);
            } while (((perf->nIterations += nSweeps) < ctl->maxIter && !check_convergence(perf, ctl)) ||
                     perf->nIterations < ctl->minIter);
        }
    }
    free(Apsi); free(tmp); free(res);
}
