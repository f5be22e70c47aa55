#!/bin/sh
# Cuts the reference's functor definitions out of the (otherwise un-compilable) .C files they live in, by line range, into a
# scratch directory OUTSIDE the repository; ref_fvm_tu.cpp #includes the fragments from there and the directory is deleted after
# the compile (oracle/Makefile).  Nothing of the reference's text is kept in the tree or shipped to the GPU box.
set -e
REF=${REF_SRC:-/root/reference/src}
OUT=$1
FV=$REF/finiteVolume
cut() { sed -n "$2,$3p" "$1" > "$OUT/$4"; grep -q "$5" "$OUT/$4" || { echo "extract.sh: '$5' not found in $1:$2-$3 (reference changed?)" >&2; exit 1; }; }
cut $FV/fvMatrices/fvMatrix/fvMatrix.C 36 76 fvMatrix_patchAdd.inc fvMatrixPatchAddFunctor
cut $FV/fvMatrices/fvMatrix/fvMatrix.C 245 286 fvMatrix_boundarySource.inc fvMatrixAddBoundarySourceFunctor
cut $FV/fvMatrices/fvMatrix/fvMatrix.C 352 452 fvMatrix_setValues.inc fvMatrixSetValuesSourceFunctor
cut $FV/fvMatrices/fvMatrix/fvMatrix.C 983 1084 fvMatrix_relax.inc fvMatrixRelaxAddToDiagonalFunctor
cut $FV/finiteVolume/fvc/fvcSurfaceIntegrate.C 40 132 fvcSurfaceIntegrate.inc surfaceIntegratePatchFunctor
cut $FV/finiteVolume/gradSchemes/gaussGrad/gaussGrad.C 31 131 gaussGrad.inc gaussGradPatchFunctor
cut $FV/interpolation/surfaceInterpolation/surfaceInterpolationScheme/surfaceInterpolationScheme.C 273 279 interpolate.inc surfaceInterpolationSchemeInterpolateFunctor
cut $FV/interpolation/surfaceInterpolation/limitedSchemes/limitedSurfaceInterpolationScheme/limitedSurfaceInterpolationScheme.C 155 161 limitedWeights.inc limitedSurfaceInterpolationSchemeWeightsFunctor
cut $FV/interpolation/surfaceInterpolation/limitedSchemes/LimitedScheme/LimitedScheme.C 32 57 calcLimiter.inc LimitedSchemeCalcLimiterFunctor
cut $REF/OpenFOAM/matrices/lduMatrix/lduMatrix/lduMatrixTemplates.C 35 50 faceH.inc lduMatrixfaceHFunctor
