// diag.hip.h -- the one switch for measurement builds of the kernels.
//
// The library the tests, bench.py and the adapters load is built WITHOUT ADSB_AMD_DIAG_BUILD: every constant below then has its product
// value, every `if (diag::k...)` in the kernels folds away and stamp() is empty.  tools/build_variant.sh makes measurement builds
// (/tmp/ab_libs/<name>.so: outside the repository, never loaded by the suite; what an A/B needs on the GPU box is copied to ab_ship/ for that call):
//     -DADSB_AMD_DIAG_BUILD=1 -DDIAG_PARTS=n        scan1090_kernel / scan2400_kernel with the later parts compiled out (tools/parts.sh;
//                                                   such a build emits no records: it is for kernel times and counters only)
//     -DADSB_AMD_DIAG_BUILD=1 -DDIAG_ORDER_PARTS=n  the same for the ordering pass
//     -DADSB_AMD_DIAG_BUILD=1 -DDIAG_STAMPS=1       every wave notes the 100 MHz clock when it comes in and goes out (tools/stamps.py)
//     -DADSB_AMD_DIAG_BUILD=1 -DDIAG_UAT=1          the UAT demodulation kernel counts what it does (tools/uat_diag.py)
//     -DADSB_AMD_DIAG_BUILD=1 -DDIAG_UAT_PARTS=n    the UAT demodulation kernel with the later parts of a match compiled out (no frames come out):
//                                                   1 staging, 2 + sync re-check, 3 + slicing, 4 + syndromes, 5 + decoding, 6 everything for ADS-B
//                                                   matches (1 - 6: uplink matches are passed over)
// Variants that were measured and rejected are not kept in the sources: their figures are in profiles/r0N_sweep.txt.
#pragma once

#if defined(ADSB_AMD_DIAG_BUILD) && ADSB_AMD_DIAG_BUILD
#ifndef DIAG_PARTS
#define DIAG_PARTS 99
#endif
#ifndef DIAG_ORDER_PARTS
#define DIAG_ORDER_PARTS 99
#endif
#ifndef DIAG_STAMPS
#define DIAG_STAMPS 0
#endif
#ifndef DIAG_UAT
#define DIAG_UAT 0
#endif
#ifndef DIAG_UAT_PARTS
#define DIAG_UAT_PARTS 99
#endif
#else
#define DIAG_PARTS 99
#define DIAG_ORDER_PARTS 99
#define DIAG_STAMPS 0
#define DIAG_UAT 0
#define DIAG_UAT_PARTS 99
#endif

namespace adsb_amd
{
namespace diag
{
constexpr int  kParts      = DIAG_PARTS;       // scan kernels: parts up to this number are compiled in (each kernel numbers its own)
constexpr int  kOrderParts = DIAG_ORDER_PARTS; // ordering pass: 1 = the prefix only, 2 = + fetch and store, 3 = + ranks, more = everything
constexpr bool kStamps     = DIAG_STAMPS != 0; // ScanArgs carries a stamp array and the kernels write to it
constexpr bool kUat        = DIAG_UAT != 0;
constexpr int  kUatParts   = DIAG_UAT_PARTS;   // UAT demodulation kernel: what is done for a match (see above)
} // namespace diag
} // namespace adsb_amd
