// ABI version + per-thread error text of libv2x_amd.so (see include/v2x_amd.h).
#include "common.h"

static thread_local char g_err[512] = "";

void v2x_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// A library that contains an INSTRUMENTED kernel object (tools/probes/*.patch built with a -DV2X_*_DBG_BUILD flag: phase-removal switches, time stamps -- its results are
// garbage) defines v2x_probe_build_marker; the version then reads NEGATIVE, which every loader that checks it refuses (v2x_sim_amd/_lib.py: unless V2X_ALLOW_PROBE_BUILD=1).
extern "C" __attribute__((weak)) int v2x_probe_build_marker;
extern "C" int v2x_abi_version(void) { return &v2x_probe_build_marker != nullptr ? -V2X_AMD_ABI_VERSION : V2X_AMD_ABI_VERSION; }

// ---- tuning switches (common.h: v2x_tune_id) -----------------------------------------------------------------------------------
#include <atomic>
#include <mutex>
#include <strings.h>
namespace {
struct TuneEntry { const char *name; int def; };
const TuneEntry kTune[V2X_TUNE_COUNT] = {
    {"STREAM_WAVES", 8}, {"STREAM_G", 1}, {"STREAM_WT", 1}, {"STORE_X4", 1}, {"STREAM_PERSIST", 1}, {"STREAM_WIDE", 1},
    {"WIDE3", 1}, {"HALO_PP", 1}, {"VOXELIZE_LDS", 1}, {"WARP_LDS", 2}, {"S2_G", 1}, {"GRU_XCD_WALK", 1}, {"HALO_XCD", 1}, {"WGRAD_TR", 1}, {"BN_PARTIAL_T", 1}, {"WGRAD_REDUCE4", 1}, {"CONV1X1", 1},
};
std::atomic<int> g_tune[V2X_TUNE_COUNT];
std::once_flag g_tune_once;
void tune_init() {
    for (int i = 0; i < V2X_TUNE_COUNT; ++i) {
        char env[64];
        snprintf(env, sizeof(env), "V2X_%s", kTune[i].name);
        const char *e = getenv(env);                       // the ONLY getenv of the library, once per process
        g_tune[i].store((e && *e) ? atoi(e) : kTune[i].def, std::memory_order_relaxed);
    }
}
int tune_find(const char *name) {
    if (!name) return -1;
    if (strncasecmp(name, "V2X_", 4) == 0) name += 4;
    for (int i = 0; i < V2X_TUNE_COUNT; ++i)
        if (strcasecmp(name, kTune[i].name) == 0) return i;
    return -1;
}
}  // namespace

int v2x_tune(int id) {
    std::call_once(g_tune_once, tune_init);
    return g_tune[id].load(std::memory_order_relaxed);
}
extern "C" int v2x_tuning_set(const char *name, int value) {
    const int i = tune_find(name);
    V2X_REQUIRE(i >= 0, "v2x_tuning_set: unknown switch '%s'", name ? name : "(null)");
    std::call_once(g_tune_once, tune_init);
    g_tune[i].store(value, std::memory_order_relaxed);
    return V2X_OK;
}
extern "C" int v2x_tuning_get(const char *name, int *value) {
    const int i = tune_find(name);
    V2X_REQUIRE(i >= 0 && value, "v2x_tuning_get: unknown switch '%s' or null result", name ? name : "(null)");
    *value = v2x_tune(i);
    return V2X_OK;
}
extern "C" const char *v2x_last_error(void) { return g_err; }
