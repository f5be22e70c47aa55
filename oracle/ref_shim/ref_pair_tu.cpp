// Translation unit that compiles the REFERENCE's pairGAMGAgglomerate.C where it lies: the shim is included first and
// owns the include guards of the two OpenFOAM headers that file asks for, so its own #include lines become no-ops.
// REF_SRC is given on the command line (oracle/Makefile).
#include "pairGAMGAgglomeration.H"   // defines pairGAMGAgglomeration_H
#ifndef lduAddressing_H
#define lduAddressing_H
#endif
#include REF_SRC
#include "ref_pair_wrapper.cpp"
