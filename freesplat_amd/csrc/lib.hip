// lib.hip -- library identification and per-thread error string of libfreesplat_hip.so.
#include <stdio.h>

#include <mutex>
#include <vector>

#include "fs_common.h"

namespace fs {
static thread_local char g_last_error[256] = "";
void set_last_error(const char* what, hipError_t e)
{
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", what, hipGetErrorString(e));
}

// ---- event-pair timing around kernel launches -------------------------------------------------
static unsigned g_profile_mask = 0;  // bit i: stage i is timed
struct Rec { int stage, units; hipEvent_t a, b; };
static std::vector<Rec> g_recs;
static std::mutex g_mu;
// at most this many launches of a stage are timed between two fs_profile_collect calls (the first ones): every timed
// launch holds two timing events until the collect, and a few thousand outstanding ones slow the submission path down
// (bench.py --steps 100: -8 % views/s with every launch timed); the per-launch average does not need more samples
constexpr int kMaxTimedPerStage = 512;
static int g_timed[kNumStages] = {};

ScopedStage::ScopedStage(Stage s, hipStream_t st, int units) : slot_(-1), st_(st)
{
    if (!((g_profile_mask >> (int)s) & 1u)) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_timed[(int)s] >= kMaxTimedPerStage) return;
    ++g_timed[(int)s];
    Rec r;
    r.stage = (int)s;
    r.units = units;
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, st);
    g_recs.push_back(r);
    slot_ = (int)g_recs.size() - 1;
}
ScopedStage::~ScopedStage()
{
    if (slot_ < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    (void)hipEventRecord(g_recs[slot_].b, st_);
}
static const char* kStageNames[kNumStages] = {"preprocess", "tile_scan", "render",
                                              "render_bwd", "preprocess_bwd", "cost_volume", "ptf", "encoder_tail"};
}  // namespace fs

FS_API int fs_profile_enable(int stage_mask)
{
    std::lock_guard<std::mutex> lk(fs::g_mu);
    fs::g_profile_mask = (unsigned)stage_mask;
    return FS_OK;
}
FS_API const char* fs_profile_stage_name(int i)
{
    return (i >= 0 && i < fs::kNumStages) ? fs::kStageNames[i] : nullptr;
}
FS_API int fs_profile_collect(int n, float* ms_total, int32_t* launches)
{
    if (n < 0 || (n > 0 && (!ms_total || !launches))) return FS_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(fs::g_mu);
    for (int i = 0; i < n; ++i) { ms_total[i] = 0.0f; launches[i] = 0; }
    int rc = FS_OK;
    for (auto& r : fs::g_recs) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            if (r.stage < n) { ms_total[r.stage] += ms; launches[r.stage] += r.units; }
        } else {
            rc = FS_ERR_LAUNCH;
        }
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    fs::g_recs.clear();
    for (int i = 0; i < fs::kNumStages; ++i) fs::g_timed[i] = 0;
    return rc;
}

FS_API const char* fs_version(void) { return "freesplat_amd 0.6.0 gfx950"; }
FS_API int fs_abi_version(void) { return FS_ABI_VERSION; }
FS_API const char* fs_last_error(void) { return fs::g_last_error; }
