// b4d_timing_only.hpp -- one umbrella for every timing-only switch of the sources.
//
// B4D_EXP_* macros build kernels that move (roughly) the right bytes but compute WRONG results; they exist to price a
// single ingredient of a kernel (twiddle loads, an LDS exchange, a store pattern) in an A/B run of tools/dev_*.py.
// They compile only together with -DB4D_TIMING_ONLY, and a library built that way says so in b4d_version() -- which
// tests/test_cabi.py refuses -- so that a wrong-result build can never pass for the product library.
#pragma once
#if !defined(B4D_TIMING_ONLY) &&                                                                                              \
    (defined(B4D_EXP_NOTW) || defined(B4D_EXP_NOXCHG) || defined(B4D_EXP_NOBAR) || defined(B4D_EXP_NOMIRROR) ||               \
     defined(B4D_EXP_ALIGNED_MIRROR) || defined(B4D_EXP_PSD_TILED) || defined(B4D_EXP_PM_SKIP23) || defined(B4D_EXP_PM_NOTAB) || \
     defined(B4D_EXP_PM_PACKED) || defined(B4D_EXP_WMR) || defined(B4D_COL_SPLIT) || defined(B4D_NO_PK_PRODUCT))
#error "B4D_EXP_* / B4D_COL_SPLIT are timing-only switches: build them with -DB4D_TIMING_ONLY (libb4d_alt.so), never into libb4d.so"
#endif
#ifdef B4D_TIMING_ONLY
#define B4D_VERSION_SUFFIX " TIMING-ONLY BUILD (B4D_EXP_* switches: results are wrong on purpose)"
#else
#define B4D_VERSION_SUFFIX ""
#endif
