// SHIM: see ref_atmul_tu.cpp
