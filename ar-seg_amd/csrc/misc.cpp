// Version / status strings of libarseg_hip.so (host only).
#include "arseg_hip.h"

extern "C" int arseg_version(void) { return ARSEG_ABI_VERSION; }

extern "C" const char *arseg_status_string(int status) {
    switch (status) {
        case ARSEG_OK: return "ok";
        case ARSEG_EINVAL: return "invalid argument (null pointer, non-positive size, misaligned pointer or stride)";
        case ARSEG_EUNSUPPORTED: return "shape not supported by the compiled kernels";
        case ARSEG_EWORKSPACE: return "workspace missing or too small";
        default: return status > 0 ? "HIP runtime error (value is the hipError_t)" : "unknown arseg status";
    }
}
