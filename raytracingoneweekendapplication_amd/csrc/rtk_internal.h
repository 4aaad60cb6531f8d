// rtk_internal.h -- what the translation units of librtk_hip.so share besides the public ABI (include/rtk.h).
#ifndef RTK_INTERNAL_H
#define RTK_INTERNAL_H

#include <hip/hip_runtime.h>

#include <string>

#include "rtk.h"

namespace rtk {

extern thread_local std::string g_error;            // text behind rtk_last_error()
int fail(int code, const char* fmt, ...);           // records the text, returns `code`

int ctx_device(const rtk_ctx* ctx);

// Block until streams[i] (on ctxs[i]'s device) has drained, i = 0..n-1, feeding ctxs[0]'s progress callback
// (rtk_set_progress_callback) from the work-item counters of the launches in flight.
int wait_with_progress(rtk_ctx* const* ctxs, const hipStream_t* streams, int n);

}  // namespace rtk

#endif  // RTK_INTERNAL_H
