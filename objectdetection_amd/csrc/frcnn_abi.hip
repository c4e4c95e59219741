// ABI bookkeeping for libfrcnn_hip.so.
#include "frcnn_common.h"

extern "C" int frcnn_abi_version(void) { return 1; }

extern "C" const char* frcnn_status_string(int status) {
    switch (status) {
        case FRCNN_OK: return "ok";
        case FRCNN_E_BADARG: return "bad argument";
        case FRCNN_E_WORKSPACE: return "workspace missing or too small";
        case FRCNN_E_LAUNCH: return "kernel launch failed";
        case FRCNN_E_UNSUPPORTED: return "size not supported by this build";
        case FRCNN_E_STATIC_LIMIT: return "needs more scratch than the built-in array holds (FRCNN_NMS_STATIC_BYTES)";
        default: return "unknown status";
    }
}
