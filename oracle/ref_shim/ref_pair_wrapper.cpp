// C entry point around the REFERENCE's pairGAMGAgglomeration::agglomerate (compiled from /root/reference, see the shim header)
#include "pairGAMGAgglomeration.H"

bool Foam::pairGAMGAgglomeration::forward_ = true;   // defined in the reference's pairGAMGAgglomeration.C:33

extern "C" int ref_pair_agglomerate(int nCells, int nFaces, const int* lower, const int* upper, const double* weights,
                                    int forward, int* coarseCellMap, int* forwardAfter)
{
    Foam::lduAddressing addr(nCells, nFaces, lower, upper);
    Foam::scalarField w(nFaces);
    for (int f = 0; f < nFaces; f++) w[f] = weights[f];
    Foam::pairGAMGAgglomeration agg;
    Foam::pairGAMGAgglomeration::forward_ = forward != 0;
    Foam::label nCoarse = -1;
    Foam::tmp<Foam::labelField> t = agg.agglomerate(nCoarse, addr, w);
    for (int c = 0; c < nCells; c++) coarseCellMap[c] = t()[c];
    if (forwardAfter) *forwardAfter = Foam::pairGAMGAgglomeration::forward_ ? 1 : 0;
    return nCoarse;
}
