// Error string, ABI version and the optional per-kernel HIP-event profiler of libmarius_hip.so.
#include <cstdlib>
#include <mutex>
#include <vector>

#include "common.h"

namespace marius {
static thread_local char g_last_error[512] = "";
void set_last_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

static KernelEnv read_kernel_env() {
    KernelEnv k = {};
    auto first = [](const char* name) -> char { const char* e = getenv(name); return e ? e[0] : (char)0; };
    k.scores = first("MARIUS_SCORES");
    k.no_fast = first("MARIUS_NO_FAST") == '1';
    k.kernels = first("MARIUS_KERNELS");
    k.no_vlog = first("MARIUS_NO_VLOG") == '1';
    k.timeline_grads = getenv("MARIUS_TIMELINE_GRADS") != nullptr;
    {
        const char* e = getenv("MARIUS_FLASH_WIDE");
        k.flash_wide = !e ? -1 : (e[0] == '0' ? 0 : atoi(e));
    }
    k.flash = first("MARIUS_FLASH");
    if (const char* e = getenv("MARIUS_FLASH_RESERVE")) { k.has_flash_reserve = true; k.flash_reserve = atoi(e); }
    if (const char* e = getenv("MARIUS_FLASH_NWG")) { k.has_flash_nwg = true; k.flash_nwg = atoi(e); }
    k.flash_f16_off = first("MARIUS_FLASH_F16") == '0';
    k.flash_rotate_off = first("MARIUS_FLASH_ROTATE") == '0';
    k.flash_tail4_off = first("MARIUS_FLASH_TAIL4") == '0';
    k.seg_fused_fixup_off = first("MARIUS_SEG_FUSED_FIXUP") == '0';
    k.seg_group_off = first("MARIUS_SEG_GROUP") == '0';
    k.sort_rocprim = first("MARIUS_SORT") == 'r';
    k.maps_fused = first("MARIUS_MAPS") == 'f';
    {
        const char* e = getenv("MARIUS_PM_NWG");
        k.pm_nwg = e ? atoi(e) : 0;
    }
    {
        const char* e = getenv("MARIUS_MT_THREADS");
        k.mt_threads = e ? atoi(e) : 0;
    }
    {
        const char* e = getenv("MARIUS_SYNC_LAUNCH");
        k.sync_launch = e ? atoi(e) : 0;
    }
    return k;
}
static KernelEnv g_kernel_env = read_kernel_env();  // at library load
const KernelEnv& kernel_env() { return g_kernel_env; }
}  // namespace marius
extern "C" int marius_config_reload(void) {
    marius::g_kernel_env = marius::read_kernel_env();
    return MARIUS_OK;
}
namespace marius {

int launch_debug(const char* what) {
    const int mode = kernel_env().sync_launch;
    if (!mode) return MARIUS_OK;
    if (mode >= 2) fprintf(stderr, "[launch] %s\n", what);
    hipError_t e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        set_last_error("%s: %s (reported by the synchronisation after the launch)", what, hipGetErrorString(e));
        fprintf(stderr, "[launch] %s FAILED: %s\n", what, hipGetErrorString(e));
        return MARIUS_ERR_HIP;
    }
    return MARIUS_OK;
}

// ---- HIP-event profiler: events are recorded on the launch stream around selected kernels (bench.py roofline) ----
static const char* kProfNames[PROF_COUNT] = {"lp_scores", "lp_grad_adj", "lp_grad_neg", "lp_prep", "lp_lse", "lp_edge_bwd",
                                             "gather_rows", "segment_adagrad_scatter", "sort_unique", "mt19937_fill", "lp_pack"};
static int g_prof_on = 0;
static std::mutex g_prof_mu;
struct ProfRec {
    int id;
    hipEvent_t a, b;
};
static std::vector<ProfRec> g_prof_pending;
static double g_prof_ms[PROF_COUNT];
static long g_prof_cnt[PROF_COUNT];

bool prof_enabled() { return g_prof_on != 0; }

void prof_begin(int id, hipStream_t st, ProfMark& m) {
    m.id = -1;
    if (!g_prof_on) return;
    if (g_prof_on >= 2 && id != g_prof_on - 2) return;  // single-kernel mode
    // device-scope release: a default event's record is a system-scope fence (L2 write-back) the bracketed kernel and its successor pay for
    if (hipEventCreateWithFlags(&m.a, hipEventReleaseToDevice) != hipSuccess || hipEventCreateWithFlags(&m.b, hipEventReleaseToDevice) != hipSuccess) return;
    m.id = id;
    (void)hipEventRecord(m.a, st);
}
void prof_end(hipStream_t st, ProfMark& m) {
    if (m.id < 0) return;
    (void)hipEventRecord(m.b, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_pending.push_back({m.id, m.a, m.b});
}
}  // namespace marius

using namespace marius;

extern "C" int marius_hip_abi_version(void) { return MARIUS_HIP_ABI_VERSION; }
extern "C" int marius_hip_struct_bytes(int which) { return which == 0 ? (int)sizeof(marius_lp_desc) : which == 1 ? (int)sizeof(marius_lp_layout) : which == 2 ? (int)sizeof(marius_segment_update) : -1; }
extern "C" const char* marius_hip_last_error(void) { return marius::g_last_error; }

extern "C" int marius_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on;
    return MARIUS_OK;
}

extern "C" int marius_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_pending) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof_pending.clear();
    for (int i = 0; i < PROF_COUNT; ++i) {
        g_prof_ms[i] = 0;
        g_prof_cnt[i] = 0;
    }
    return MARIUS_OK;
}

extern "C" int marius_profile_kernel_count(void) { return PROF_COUNT; }
extern "C" const char* marius_profile_kernel_name(int id) { return (id >= 0 && id < PROF_COUNT) ? kProfNames[id] : ""; }

// Waits for the recorded events, folds them into the per-kernel totals and returns total ms / launches of kernel `id`.
extern "C" int marius_profile_read(int id, double* total_ms, int64_t* launches) {
    MARIUS_REQUIRE(id >= 0 && id < PROF_COUNT && total_ms && launches, "profile_read: bad arguments");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    for (auto& r : g_prof_pending) {
        (void)hipEventSynchronize(r.b);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            g_prof_ms[r.id] += ms;
            g_prof_cnt[r.id] += 1;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    g_prof_pending.clear();
    *total_ms = g_prof_ms[id];
    *launches = g_prof_cnt[id];
    return MARIUS_OK;
}
