// SHIM: see ref_gamg_scale_tu.cpp
