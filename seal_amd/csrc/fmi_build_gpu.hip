// Device-side index construction (suffix array by prefix doubling with rocPRIM
// radix sorts, BWT, wavelet-matrix levels).  Placeholder until the GPU builder
// lands: reports FMI_ERR_UNSUPPORTED so callers fall back to fmi_build (host).
#include <hip/hip_runtime.h>
#include "fmi_internal.h"

extern "C" int fmi_build_device(fmi_t *h, const uint32_t *d_data, uint64_t n_data, int device, int keep_host)
{
    (void)h; (void)d_data; (void)n_data; (void)device; (void)keep_host;
    fmi_set_error("fmi_build_device: not available in this build");
    return FMI_ERR_UNSUPPORTED;
}
