// ABI bookkeeping for libfrcnn_hip.so: version, status strings, tuning values.
#include "frcnn_common.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

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

// ---- tuning values: the only mutable process-wide state besides frcnn_nms_static's array.
// The environment is read once, by the static initialiser below (i.e. at dlopen), so no entry
// point ever calls getenv(): a host thread changing the environment later cannot race a call,
// and a HIP graph capture and a later eager call select the same kernels.
namespace {
struct TuneName { const char* key; const char* env; };
const TuneName kTune[frcnn::TUNE_COUNT] = {
    {"nms_gk", "FRCNN_NMS_GK"},
    {"roi_band_min_wg", "FRCNN_ROI_BAND_MIN_WG"},
    {"bwd_cx", "FRCNN_BWD_CX"},
    {"bwd_chunk", "FRCNN_BWD_CHUNK"},
    {"bwd_band", "FRCNN_BWD_BAND"},
    {"nms_fused_max_b", "FRCNN_NMS_FUSED_MAX_B"},
    {"roi_band_split", "FRCNN_ROI_BAND_SPLIT"},
    {"topk_merge_max_runs", "FRCNN_TOPK_MERGE_MAX_RUNS"},
};
std::atomic<long long> g_tune[frcnn::TUNE_COUNT];
struct TuneInit {
    TuneInit() {
        for (int i = 0; i < frcnn::TUNE_COUNT; ++i) {
            const char* e = getenv(kTune[i].env);
            g_tune[i].store(e && *e ? atoll(e) : frcnn::TUNE_UNSET, std::memory_order_relaxed);
        }
    }
} g_tune_init;
int tune_index(const char* key) {
    if (!key) return -1;
    for (int i = 0; i < frcnn::TUNE_COUNT; ++i)
        if (!strcmp(key, kTune[i].key) || !strcmp(key, kTune[i].env)) return i;
    return -1;
}
}  // namespace

long long frcnn::tuning(frcnn::Tuning t) { return g_tune[t].load(std::memory_order_relaxed); }

extern "C" int frcnn_set_tuning(const char* key, long long value) {
    const int i = tune_index(key);
    if (i < 0) return FRCNN_E_BADARG;
    g_tune[i].store(value, std::memory_order_relaxed);
    return FRCNN_OK;
}

extern "C" int frcnn_get_tuning(const char* key, long long* value) {
    const int i = tune_index(key);
    if (i < 0 || !value) return FRCNN_E_BADARG;
    *value = g_tune[i].load(std::memory_order_relaxed);
    return FRCNN_OK;
}
