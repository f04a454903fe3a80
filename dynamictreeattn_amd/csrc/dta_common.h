// Shared host-side launch bookkeeping of the C ABI (include/dta.h).
#ifndef DTA_COMMON_H
#define DTA_COMMON_H
#include <hip/hip_runtime.h>
#include "../../include/dta.h"

// A HIP error that is ALREADY pending when an entry point is called (an earlier asynchronous kernel fault, or a failed
// runtime call nobody checked) must neither be swallowed by this launch nor be blamed on it: nothing is launched and the
// caller gets DTA_EPRIOR; dta_take_pending_error() names and clears it.  (Round 1 cleared it silently here.)
#define DTA_REFUSE_IF_PRIOR_ERROR() do { if (hipPeekAtLastError() != hipSuccess) return DTA_EPRIOR; } while (0)
// status of THIS launch (configuration errors: invalid grid, too much LDS, no code object for the device ...)
#define DTA_LAUNCH_STATUS() (hipGetLastError() == hipSuccess ? DTA_OK : DTA_ELAUNCH)

// fp32 tree attention (tree_attn_f32.hip): reached through dta_tree_attn_fwd_ex / dta_tree_attn_bwd_ex with dtype DTA_F32
int dta_attn_fwd_f32(const void* q, const void* k, const void* v, void* out, float* lse, const int32_t* subtree_end, const int32_t* run_ptr,
                     const int32_t* runs, int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv,
                     int64_t q_st, int64_t q_sh, int64_t kv_st, int64_t kv_sh, int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh,
                     float scale, hipStream_t st);
int dta_attn_bwd_f32(const void* q, const void* k, const void* v, const void* out, const void* dout, const float* lse, float* delta,
                     void* dq, void* dk, void* dv, const int32_t* subtree_end, const int32_t* run_ptr, const int32_t* runs, const int32_t* ktile_qend,
                     int32_t Tq, int32_t Tk, int32_t q_offset, int32_t Hq, int32_t Hkv,
                     int64_t q_st, int64_t q_sh, int64_t kv_st, int64_t kv_sh, int64_t v_st, int64_t v_sh, int64_t o_st, int64_t o_sh,
                     int64_t dq_st, int64_t dq_sh, int64_t dkv_st, int64_t dkv_sh, float scale, int32_t accumulate, int32_t which, hipStream_t st);

#endif
