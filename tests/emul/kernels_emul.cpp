// TEST INFRASTRUCTURE: harness of the CPU build (tests/emul/build_emul.py appends one #include per transformed kernel file): the three entry points of
// error.hip the bindings expect of any libmarius_hip.
#include <algorithm>
#include <numeric>

#include "common.h"

extern "C" const char* marius_hip_last_error(void) { return marius::g_last_error; }
extern "C" int marius_hip_abi_version(void) { return MARIUS_HIP_ABI_VERSION; }
extern "C" int marius_config_reload(void) {
    marius::g_env = marius::read_env();
    return MARIUS_OK;
}

// lp_decoder.hip dispatches to the tuned kernel files when its level selects them; the CPU build stays at the generic level, where none of these is
// reached — they only have to link
#include "lp_common.h"
namespace marius {
bool launch_scores_fast(const ScoreArgs&, bool, hipStream_t) { return false; }
bool launch_grad_adj_fast(const GradArgs&, bool, hipStream_t) { return false; }
bool launch_grad_neg_fast(const GradArgs&, bool, hipStream_t) { return false; }
bool scores_res_applicable(const float*, int64_t, int) { return false; }
bool scores_a_applicable(const float*, int64_t, int) { return false; }
bool launch_scores_ap(const ScoreArgs&, bool, hipStream_t) { return false; }
bool launch_scores_res(const ScoreArgs&, bool, hipStream_t) { return false; }
bool launch_grad16(const GradArgs&, bool, int, hipStream_t) { return false; }
bool flash_applicable(const marius_lp_desc*, const LpDims&) { return false; }
size_t flash_adjrec_bytes(const LpDims&) { return 0; }
size_t flash_negrec_bytes(const LpDims&) { return 0; }
size_t flash_part_bytes(const LpDims&) { return 0; }
const float* flash_part_weights(const LpDims&, const float2*) { return nullptr; }
bool flash_fused() { return false; }
int flash_forward(const marius_lp_desc*, const LpDims&, const float*, char*, char*, float2*, float*, bool, float*, const int64_t[2], float*, const float*, float*, float*, hipStream_t) { return MARIUS_ERR_UNSUPPORTED; }
int flash_merge(const LpDims&, const float2*, const float*, float*, float*, float*, float*, char*, bool, hipStream_t) { return MARIUS_ERR_UNSUPPORTED; }
int flash_backward(const marius_lp_desc*, const LpDims&, char*, char*, float*, float*, const int64_t[2], const float2*, bool, float*, hipStream_t) { return MARIUS_ERR_UNSUPPORTED; }
int flash_chunks(int) { return 1; }
bool flash_chunked(int) { return false; }
bool flash_tail4(int) { return false; }
size_t flash_tiled_scores_bytes(const LpDims&) { return 0; }
FlRange flash_range(const marius_lp_desc*, const LpDims&) { return FlRange{nullptr, nullptr, FL_ADJ_NODE}; }
}  // namespace marius

