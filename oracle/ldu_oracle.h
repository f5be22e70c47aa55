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

#endif
