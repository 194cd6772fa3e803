// A/B and ablation switches exist only in the diagnostic build (`python km-bart_amd/build.py --variant diag KMB_DIAG`,
// selected at run time with KMB_LIB_PATH): the product library reads no such environment variable and carries no
// "skip the work" path.  Product-build knobs (plain getenv, few on purpose): KMB_FUSED_CE, KMB_GEN_FUSED, KMB_FP32_HEAD,
// KMB_NO_SIDE_STREAM, KMB_SMALL_SPLIT, KMB_WGRAD_GROUP (tests compare the two paths), KMB_GEMM_VARIANT / KMB_GEMM_AUTOTUNE /
// KMB_GEMM_TUNE_FILE (bit-identical variants; the tune file keeps tuning launches out of profiled runs).
#pragma once
#include <cstdlib>
#ifdef KMB_DIAG
#define KMB_DIAG_ENV(name) getenv(name)
#define KMB_DIAG_BIT(word, bit) (((word) & (bit)) != 0)
#else
#define KMB_DIAG_ENV(name) (static_cast<const char*>(nullptr))
#define KMB_DIAG_BIT(word, bit) false
#endif
