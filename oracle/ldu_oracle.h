/* ldu_oracle.h -- data model of the CPU ORACLE (test infrastructure, see ldu_oracle.c), shared by its files. */
#ifndef LDU_ORACLE_H
#define LDU_ORACLE_H
#include <stdint.h>
typedef int32_t label;
typedef double scalar;

typedef struct {
    label nbrDomain;   /* domain on the other side                         */
    label nbrPatch;    /* index of the matching interface in that domain    */
    label nFaces;
    label *faceCells;  /* [nFaces] local cell touched by each patch face    */
    scalar *bouCoeffs; /* interfaceBouCoeffs (used by Amul)                 */
    scalar *intCoeffs; /* interfaceIntCoeffs (used by Tmul)                 */
    /* cyclicAMI (cyclicAMIFvPatchField.C:195-224): neighbour value of face i = sum over k in [amiStart[i], amiStart[i+1]) of
     * amiW[k] * (factor * psi_nbr[nbrFaceCells[amiAddr[k]]]); amiStart NULL = ordinary one-to-one interface.  amiLow[i] != 0:
     * the face's weight sum is under lowWeightCorrection, its value is the face's own cell (the `pif` default).        */
    label *amiStart, *amiAddr;
    scalar *amiW;
    unsigned char *amiLow;
    scalar *amiMagSf;  /* [nFaces] face areas of this side (srcMagSf / tgtMagSf): only the GAMG agglomeration of the AMI reads them */
    /* cyclicAMI whose partner SIDE is split over several domains (the reference's distributed AMI, singlePatchProc_ == -1,
     * AMIInterpolation.C:940-1091: the faces a rank needs arrive from all ranks and are numbered [from rank 0][from rank 1]...):
     * nAmiParts > 0: amiAddr numbers the concatenation of the faces of interfaces (amiPartDomain[q], amiPartPatch[q]), q ascending;
     * partner q covers [amiPartStart[q], amiPartStart[q+1]).  nAmiParts == 0: the one partner (nbrDomain, nbrPatch).            */
    int nAmiParts; label *amiPartDomain, *amiPartPatch, *amiPartStart;
    scalar factor;     /* transformCoupleField: pow(diag(forwardT).component(cmpt), rank); 1 = no transformation */
} orc_iface;

typedef struct {
    label nCells, nFaces;
    label *lower, *upper;             /* lowerAddr (owner) / upperAddr (neighbour) */
    label *losort, *ownerStart, *losortStart;
    scalar *diag, *lowerC, *upperC;   /* lowerC == upperC when symmetric   */
    int symmetric;
    int nIfaces;
    orc_iface *ifaces;
    int64_t offset;                   /* start of this domain in a system vector */
} orc_domain;

typedef struct {
    int nDomains;
    orc_domain *dom;
    int64_t nTotal;
    int accurate_sums; /* 1: long double reductions (parity), 0: plain double (timing) */
} orc_system;

/* (domain, interface, face) behind address j of a cyclicAMI interface */
static inline const orc_iface *orc_ami_partner(const orc_system *s, const orc_iface *me, label j, int *domOut, label *faceOut)
{
    if (me->nAmiParts == 0) { *domOut = me->nbrDomain; *faceOut = j; return &s->dom[me->nbrDomain].ifaces[me->nbrPatch]; }
    int q = 0;
    while (q + 1 < me->nAmiParts && j >= me->amiPartStart[q + 1]) q++;
    *domOut = me->amiPartDomain[q]; *faceOut = j - me->amiPartStart[q];
    return &s->dom[me->amiPartDomain[q]].ifaces[me->amiPartPatch[q]];
}

#endif
