// Per-kernel launch timing with HIP events recorded on the stream each kernel is launched on.
// Off by default (zero overhead besides one branch); bench.py switches it on for a dedicated
// measurement pass so that roofline.achieved comes from the kernel's own launch durations.
#include "common.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace {
struct Rec { hipEvent_t a, b; int kid; double flops, bytes; };
std::mutex g_mu;
std::vector<Rec> g_open;       // begun, not yet ended (per stream nesting is not used)
std::vector<Rec> g_done;
bool g_on = false;
const char* kNames[KID_COUNT] = {
    "msda_forward_vec8", "msda_forward_generic", "msda_fused_forward", "gemm_mfma", "gemm_mfma_conv3x3",
    "gemm_mfma_patch", "attn_window", "attn_global", "attn_decoder", "layernorm_rows", "eltwise", "mlp_fused", "vit_block", "row_chain"};
}  // namespace

void lwdetr_prof_begin(int kid, double flops, double bytes, hipStream_t s) {
    if (!g_on) return;
    Rec r; r.kid = kid; r.flops = flops; r.bytes = bytes;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, s);
    std::lock_guard<std::mutex> lk(g_mu);
    g_open.push_back(r);
}

void lwdetr_prof_end(hipStream_t s) {
    if (!g_on) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_open.empty()) return;
    Rec r = g_open.back();
    g_open.pop_back();
    (void)hipEventRecord(r.b, s);
    g_done.push_back(r);
}

extern "C" {

int lwdetr_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_on = on != 0;
    return LWDETR_OK;
}

int lwdetr_prof_num_kernels() { return KID_COUNT; }

const char* lwdetr_prof_kernel_name(int kid) { return (kid >= 0 && kid < KID_COUNT) ? kNames[kid] : ""; }

// Synchronises all recorded events, accumulates per kernel id and clears the log.
// ms / flops / bytes / count are caller arrays of n >= lwdetr_prof_num_kernels() entries (accumulated into).
int lwdetr_prof_collect(double* ms, double* flops, double* bytes, long long* count, int n) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (n < KID_COUNT) return LWDETR_ERR_BAD_ARG;
    for (Rec& r : g_done) {
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms[r.kid] += t; flops[r.kid] += r.flops; bytes[r.kid] += r.bytes; count[r.kid] += 1;
        }
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_done.clear();
    return LWDETR_OK;
}

}  // extern "C"

// ---- the knob table (common.h): environment variables LWDETR_<NAME>, read once
namespace {
const char* kKnobNames[KNOB_COUNT] = {
    "ATTN_LDS_CFG", "ATTN_LDS", "ATTN_SHORT", "ATTN_WTILE", "ATTN_WIN", "ATTN_QT", "CHAIN_SPLIT_ROWS", "GEMM_BIG", "GEMM_BIG_BN", "GEMM_BIG_2WG",
    "CONV_PATCH", "GEMM_TILE", "GEMM_DMA", "GEMM_KB", "GEMM_NST", "GEMM_PT", "GEMM_PT_SKEW", "MLP_SMALL_TT", "FFN_SPLITS", "MLP_SMALL", "VB_GRID",
    "VB_GELU16", "VB_HALF", "GEMM_FEW_WAVES"};
struct KnobTable {
    long val[KNOB_COUNT]; bool set[KNOB_COUNT];
    KnobTable() {
        for (int i = 0; i < KNOB_COUNT; ++i) {
            char name[64];
            snprintf(name, sizeof(name), "LWDETR_%s", kKnobNames[i]);
            const char* e = getenv(name);
            set[i] = e != nullptr && *e != 0;
            // VB_GRID=rounds (whole rounds of one workgroup per CU) is stored as -1
            val[i] = !set[i] ? 0 : (i == KNOB_VB_GRID && !strcmp(e, "rounds") ? -1 : atol(e));
        }
    }
};
KnobTable& knobs() { static KnobTable t; return t; }
}  // namespace

long lwdetr_knob(int id, long dflt) { const KnobTable& t = knobs(); return id >= 0 && id < KNOB_COUNT && t.set[id] ? t.val[id] : dflt; }
bool lwdetr_knob_is_set(int id) { return id >= 0 && id < KNOB_COUNT && knobs().set[id]; }

extern "C" int lwdetr_tuning_set(const char* name, long value, int is_set) {
    if (!name) return LWDETR_ERR_BAD_ARG;
    if (!strncmp(name, "LWDETR_", 7)) name += 7;
    for (int i = 0; i < KNOB_COUNT; ++i)
        if (!strcmp(name, kKnobNames[i])) { KnobTable& t = knobs(); t.val[i] = value; t.set[i] = is_set != 0; return LWDETR_OK; }
    return LWDETR_ERR_BAD_ARG;
}
