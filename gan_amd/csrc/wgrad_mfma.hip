// bf16 MFMA weight-gradient kernel (placeholder dispatch until the transposed-read kernel lands).
#include "common.h"
int ga_wgrad_mfma_supported(const WgradParams& p) { (void)p; return 0; }
int ga_wgrad_mfma_impl(const WgradParams& p, hipStream_t s) { (void)p; (void)s; return GA_ERR_ARG; }
