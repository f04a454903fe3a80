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

#endif
