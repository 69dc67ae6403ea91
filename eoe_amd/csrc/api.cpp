// error state, version and kernel-timing registry of the C ABI (include/eoe_hip.h)
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>
#include "../../include/eoe_hip.h"

thread_local char g_eoe_err[512] = {0};
int eoe_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_eoe_err, sizeof(g_eoe_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int eoe_abi_version(void) { return EOE_ABI_VERSION; }

// sizeof of the argument structs as this library was compiled: a binding checks its own mirror of the layout against it
extern "C" int eoe_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(eoe_gemm_args);
        case 1: return (int)sizeof(eoe_conv_geometry);
        case 2: return (int)sizeof(eoe_adam_chunk);
        case 3: return (int)sizeof(eoe_adam_scalars);
        case 4: return (int)sizeof(eoe_vit_block_fwd_args);
        case 5: return (int)sizeof(eoe_vit_block_bwd_args);
        case 6: return (int)sizeof(eoe_cgate_args);
        case 7: return (int)sizeof(eoe_cgate_bwd_args);
        case 8: return (int)sizeof(eoe_sgate_args);
        case 9: return (int)sizeof(eoe_sgate_bwd_args);
        case 10: return (int)sizeof(eoe_adam_tile);
        case 11: return (int)sizeof(eoe_red_table);
        default: return -1;
    }
}
extern "C" const char* eoe_last_error(void) { return g_eoe_err; }

namespace {
struct Rec { char name[32]; hipEvent_t a, b; double flops, bytes; };      // the name is copied: callers may build it on their stack
std::mutex g_mu;
bool g_on = false;
std::vector<Rec> g_recs;
}  // namespace

bool eoe_prof_active() { return g_on; }

int eoe_prof_begin(const char* name, double flops, double bytes, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    Rec r{};
    strncpy(r.name, name, sizeof(r.name) - 1);
    r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return -1;
    (void)hipEventRecord(r.a, s);
    g_recs.push_back(r);
    return (int)g_recs.size() - 1;
}

void eoe_prof_finish(int idx, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (idx >= 0 && idx < (int)g_recs.size()) (void)hipEventRecord(g_recs[idx].b, s);
}

extern "C" int eoe_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (on) {
        for (auto& r : g_recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
        g_recs.clear();
    }
    g_on = on != 0;
    return 0;
}

extern "C" int eoe_prof_collect(eoe_prof_entry* out, int max_entries, int* n_out) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!out || !n_out) return eoe_set_error(EOE_ERR_ARG, "prof_collect: null pointer");
    std::map<std::string, eoe_prof_entry> agg;
    for (auto& r : g_recs) {
        if (hipEventSynchronize(r.b) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "prof_collect: event sync failed");
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) return eoe_set_error(EOE_ERR_LAUNCH, "prof_collect: elapsed failed");
        auto& e = agg[r.name];
        if (e.launches == 0) { memset(&e, 0, sizeof(e)); strncpy(e.name, r.name, sizeof(e.name) - 1); }
        e.launches += 1; e.total_ms += ms; e.flops += r.flops; e.bytes += r.bytes;
    }
    int n = 0;
    for (auto& kv : agg) { if (n < max_entries) out[n++] = kv.second; }
    *n_out = n;
    return 0;
}
