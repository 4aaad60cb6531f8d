// rtk_multi.cpp -- several GPUs of one node behind ONE call (rtk_init_multi / rtk_render_multi, include/rtk.h).
//
// The reference's camera::render owns all of its parallelism: it cuts the image into row blocks and hands them to
// std::async workers (Camera.txt:59-61,96-100).  This is that on MI355X GPUs, driven by one host thread:
//
//   * the scene is replicated (<= 1.3 MB); rtk_multi_scene_upload_fast optimises the visiting order ONCE on the host
//     and uploads the result to every device;
//   * device i renders the interleaved 8x8 tiles t with t % n == i into a compact buffer [tiles_per_rank][3][64] on
//     its own stream (rtk_render_device with rank = i, n_ranks = n) -- no exchange while rendering;
//   * ONE gather to the first device: ncclGather over xGMI (rccl.h:745; one communicator per device from
//     ncclCommInitAll, the n calls grouped), or -- when a device is listed twice, or librccl cannot be loaded -- one
//     hipMemcpyPeerAsync per device, enqueued on the PRODUCING device's stream so that each link carries its share as
//     soon as its rank has finished;
//   * rtk_tiles_unpermute on the first device writes the row-major image and the gamma / clamp / quantised bytes;
//   * rtk_render_multi_enqueue / rtk_multi_wait make that asynchronous with two frames in flight: gather and un-permute
//     run on per-device transfer streams and on a second set of tile buffers while the next frame renders (an animation
//     loop or a sequence of camera::render calls then gets what bench.py's pipelined RCCL ranks get).
//
// librccl.so is loaded with dlopen on first use: librtk_hip.so itself does not link it, single-GPU users never pay for it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "rtk.h"
#include "rtk_internal.h"

using rtk::fail;

namespace {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load() {
        if (handle) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) return false;
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(handle, "ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(handle, "ncclCommDestroy"));
        Gather = reinterpret_cast<decltype(Gather)>(dlsym(handle, "ncclGather"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(handle, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(handle, "ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(handle, "ncclGetErrorString"));
        return CommInitAll && CommDestroy && Gather && GroupStart && GroupEnd && GetErrorString;
    }
};

RcclApi g_rccl;

}  // namespace

// Frames in flight (rtk_render_multi_enqueue): frame k uses buffer set k % kSlots -- its own compact tile buffers and its own
// gather target -- so that the gather and the un-permute of frame k run while frame k + 1 renders.
constexpr int kSlots = 2;

struct rtk_multi {
    std::vector<int> devices;
    std::vector<rtk_ctx*> ctxs;
    std::vector<hipStream_t> streams;  // per device: the render stream ...
    std::vector<hipStream_t> xfer;     // ... and the stream its share of the gather runs on (device 0: also the un-permute)
    std::vector<hipEvent_t> rendered[kSlots];  // device i has rendered its tiles of the frame in this slot (render stream)
    std::vector<hipEvent_t> arrived[kSlots];   // device i's tiles of that frame have arrived on the first device (xfer stream)
    hipEvent_t released[kSlots] = {nullptr, nullptr};  // the frame in this slot has been un-permuted: its buffers may be rendered into again
    bool slot_used[kSlots] = {false, false};
    std::vector<ncclComm_t> comms;    // empty: peer copies
    // per device and slot: compact tile buffers (device 0 renders straight into its part of `gathered`); the gather targets
    // and the image buffers of rtk_render_multi on the first device; all grown on demand
    std::vector<void*> compact[kSlots];
    std::vector<size_t> compact_bytes[kSlots];
    void* gathered[kSlots] = {nullptr, nullptr};
    size_t gathered_bytes[kSlots] = {0, 0};
    void* image = nullptr;
    size_t image_bytes = 0;
    uint8_t* rgb8 = nullptr;
    size_t rgb8_bytes = 0;
    uint64_t frames_enqueued = 0;
};

namespace {

#define RTKM_HIP(call)                                                                                \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess) return fail(RTK_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

int grow(void** p, size_t* have, size_t need) {
    if (need <= *have) return RTK_OK;
    if (*p) {
        RTKM_HIP(hipDeviceSynchronize());
        RTKM_HIP(hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    RTKM_HIP(hipMalloc(p, need));
    *have = need;
    return RTK_OK;
}

}  // namespace

extern "C" {

int rtk_init_multi(int n_devices, const int* devices, int gather_mode, rtk_multi** out_multi) {
    if (!out_multi) return fail(RTK_ERR_INVALID, "rtk_init_multi: out_multi is null");
    *out_multi = nullptr;
    if (n_devices < 1 || n_devices > 64 || !devices) return fail(RTK_ERR_INVALID, "rtk_init_multi: need 1..64 devices");
    if (gather_mode < RTK_GATHER_AUTO || gather_mode > RTK_GATHER_RCCL) return fail(RTK_ERR_INVALID, "rtk_init_multi: unknown gather mode %d", gather_mode);
    auto* m = new rtk_multi;
    auto bail = [&](int rc) {
        const std::string keep = rtk::g_error;  // rtk_multi_destroy must not hide the cause
        rtk_multi_destroy(m);
        rtk::g_error = keep;
        return rc;
    };
    for (int i = 0; i < n_devices; i++) {
        rtk_ctx* ctx = nullptr;
        const int rc = rtk_init(devices[i], &ctx);  // RTK_ERR_NO_DEVICE without a gfx950 device: there is no CPU path
        if (rc != RTK_OK) return bail(rc);
        m->devices.push_back(devices[i]);
        m->ctxs.push_back(ctx);
        hipStream_t st = nullptr, xf = nullptr;
        hipError_t e = hipSetDevice(devices[i]);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) m->streams.push_back(st);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&xf, hipStreamNonBlocking);
        if (e == hipSuccess) m->xfer.push_back(xf);
        for (int s = 0; s < kSlots && e == hipSuccess; s++) {
            hipEvent_t ev = nullptr;
            e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) m->rendered[s].push_back(ev);
            ev = nullptr;
            if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e == hipSuccess) m->arrived[s].push_back(ev);
            if (i == 0 && e == hipSuccess) e = hipEventCreateWithFlags(&m->released[s], hipEventDisableTiming);
            m->compact[s].push_back(nullptr);
            m->compact_bytes[s].push_back(0);
        }
        if (e != hipSuccess) return bail(fail(RTK_ERR_HIP, "rtk_init_multi: stream/event creation on device %d failed: %s", devices[i], hipGetErrorString(e)));
    }
    // peer access towards the first device (the copies / RCCL use it where the topology offers it; failure is not fatal:
    // hipMemcpyPeerAsync then stages through the host)
    for (int i = 1; i < n_devices; i++) {
        if (devices[i] == devices[0]) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, devices[i], devices[0]) == hipSuccess && can) {
            (void)hipSetDevice(devices[i]);
            const hipError_t e = hipDeviceEnablePeerAccess(devices[0], 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
        }
    }
    const bool distinct = std::set<int>(m->devices.begin(), m->devices.end()).size() == m->devices.size();
    const char* forced = getenv("RTK_MULTI_GATHER");  // "peer" / "rccl": overrides RTK_GATHER_AUTO (diagnostics)
    if (gather_mode == RTK_GATHER_AUTO && forced) gather_mode = std::strcmp(forced, "rccl") == 0 ? RTK_GATHER_RCCL : (std::strcmp(forced, "peer") == 0 ? RTK_GATHER_PEER : RTK_GATHER_AUTO);
    const bool want_rccl = gather_mode == RTK_GATHER_RCCL || (gather_mode == RTK_GATHER_AUTO && distinct && n_devices > 1);
    if (want_rccl) {
        const char* why = nullptr;
        if (!distinct) why = "a device is listed more than once";
        else if (!g_rccl.load()) why = "librccl.so could not be loaded";
        else {
            m->comms.assign(size_t(n_devices), nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(m->comms.data(), n_devices, m->devices.data());
            if (r != ncclSuccess) {
                why = g_rccl.GetErrorString(r);
                m->comms.clear();
            }
        }
        if (why && gather_mode == RTK_GATHER_RCCL) return bail(fail(RTK_ERR_UNSUPPORTED, "rtk_init_multi: RCCL gather requested but %s", why));
    }
    *out_multi = m;
    return RTK_OK;
}

int rtk_multi_destroy(rtk_multi* m) {
    if (!m) return RTK_OK;
    for (ncclComm_t c : m->comms)
        if (c) (void)g_rccl.CommDestroy(c);
    for (size_t i = 0; i < m->ctxs.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        for (auto* list : {&m->streams, &m->xfer})
            if (i < list->size() && (*list)[i]) (void)hipStreamSynchronize((*list)[i]);
    }
    for (size_t i = 0; i < m->ctxs.size(); i++) {
        (void)hipSetDevice(m->devices[i]);
        for (auto* list : {&m->streams, &m->xfer})
            if (i < list->size() && (*list)[i]) (void)hipStreamDestroy((*list)[i]);
        for (int s = 0; s < kSlots; s++) {
            if (i < m->rendered[s].size() && m->rendered[s][i]) (void)hipEventDestroy(m->rendered[s][i]);
            if (i < m->arrived[s].size() && m->arrived[s][i]) (void)hipEventDestroy(m->arrived[s][i]);
            if (i < m->compact[s].size() && m->compact[s][i]) (void)hipFree(m->compact[s][i]);
        }
    }
    if (!m->devices.empty()) {
        (void)hipSetDevice(m->devices[0]);
        for (int s = 0; s < kSlots; s++) {
            if (m->released[s]) (void)hipEventDestroy(m->released[s]);
            if (m->gathered[s]) (void)hipFree(m->gathered[s]);
        }
        if (m->image) (void)hipFree(m->image);
        if (m->rgb8) (void)hipFree(m->rgb8);
    }
    for (rtk_ctx* c : m->ctxs) rtk_destroy(c);
    delete m;
    return RTK_OK;
}

int rtk_multi_device_count(const rtk_multi* m) { return m ? int(m->ctxs.size()) : 0; }

int rtk_multi_uses_rccl(const rtk_multi* m) { return (m && !m->comms.empty()) ? 1 : 0; }

rtk_ctx* rtk_multi_ctx(rtk_multi* m, int i) { return (m && i >= 0 && i < int(m->ctxs.size())) ? m->ctxs[size_t(i)] : nullptr; }

int rtk_multi_scene_upload(rtk_multi* m, const rtk_scene_desc* scene) {
    if (!m || !scene) return fail(RTK_ERR_INVALID, "rtk_multi_scene_upload: null argument");
    for (rtk_ctx* c : m->ctxs) {
        const int rc = rtk_scene_upload(c, scene);
        if (rc != RTK_OK) return rc;
    }
    return RTK_OK;
}

int rtk_multi_scene_upload_fast(rtk_multi* m, const rtk_scene_desc* scene, const rtk_optimize_opts* opts, rtk_optimize_info* info) {
    if (!m || !scene) return fail(RTK_ERR_INVALID, "rtk_multi_scene_upload_fast: null argument");
    rtk_scene_desc* fast = nullptr;
    int rc = rtk_scene_optimize(scene, opts, &fast, info);  // once, on the host: the hierarchy does not depend on the device
    if (rc != RTK_OK) return fail(rc, "rtk_multi_scene_upload_fast: rtk_scene_optimize rejected the scene description");
    for (rtk_ctx* c : m->ctxs) {
        rc = rtk_scene_upload_optimized(c, fast, opts);
        if (rc != RTK_OK) break;
    }
    rtk_scene_optimized_free(fast);
    return rc;
}

// Which buffer set frame `frame` (0, 1, 2, ...) uses and what its renders must wait for: host-only, the rule
// rtk_render_multi_enqueue follows (exported so that the ordering can be tested where there is no GPU).
// out[0] = slot, out[1] = 1 when the renders must first wait for the release of that slot by frame `frame - kSlots`.
int rtk_multi_frame_plan(int64_t frame, int32_t out[2]) {
    if (frame < 0 || !out) return fail(RTK_ERR_INVALID, "rtk_multi_frame_plan: bad argument");
    out[0] = int32_t(frame % kSlots);
    out[1] = frame >= kSlots ? 1 : 0;
    return RTK_OK;
}

int rtk_render_multi_enqueue(rtk_multi* m, const rtk_camera* cam, const rtk_render_opts* opts, void* d_linear, uint8_t* d_rgb8) {
    if (!m || !cam || !opts) return fail(RTK_ERR_INVALID, "rtk_render_multi_enqueue: null argument");
    if (opts->count_work) return fail(RTK_ERR_INVALID, "rtk_render_multi_enqueue: work counters are per device (use rtk_render_device)");
    if (opts->real_mode != RTK_REAL_F64 && opts->real_mode != RTK_REAL_F32) return fail(RTK_ERR_INVALID, "rtk_render_multi_enqueue: unknown real_mode %d", opts->real_mode);
    const int n = int(m->ctxs.size());
    if (n == 1 && m->comms.empty()) {  // one device: no tile buffers and no gather, the image directly (with RTK_GATHER_RCCL forced, one
                                       // device still goes through a 1-rank ncclGather: that is how the RCCL path is tested on a 1-GPU box)
        rtk_render_opts o = *opts;
        o.rank = 0;
        o.n_ranks = 1;
        o.stream = m->streams[0];
        const int rc1 = rtk_render_device(m->ctxs[0], cam, &o, d_linear, d_rgb8, nullptr);
        if (rc1 == RTK_OK) m->frames_enqueued++;
        return rc1;
    }
    const size_t elem = opts->real_mode == RTK_REAL_F64 ? sizeof(double) : sizeof(float);
    const int64_t tpr = rtk_tiles_per_rank(cam->image_width, cam->image_height, n);
    if (tpr <= 0) return fail(RTK_ERR_INVALID, "rtk_render_multi_enqueue: bad image size");
    const size_t part = size_t(tpr) * 3 * RTK_TILE_PIXELS * elem;  // one device's compact buffer
    int32_t plan[2];
    (void)rtk_multi_frame_plan(int64_t(m->frames_enqueued), plan);
    const int slot = plan[0];
    // a buffer of this set has to grow (first use, or a larger frame): nothing may still be reading or filling the old one --
    // peer copies and gathers of frames in flight run on OTHER devices' streams, which a device-local synchronise does not see
    bool must_grow = part * size_t(n) > m->gathered_bytes[slot];
    for (int i = 1; i < n; i++) must_grow = must_grow || part > m->compact_bytes[slot][size_t(i)];
    int rc = RTK_OK;
    if (must_grow && m->frames_enqueued > 0) {  // (plain synchronisation: no progress callback for frames that are not this call's)
        for (int i = 0; i < n; i++) {
            RTKM_HIP(hipSetDevice(m->devices[size_t(i)]));
            RTKM_HIP(hipStreamSynchronize(m->streams[size_t(i)]));
            RTKM_HIP(hipStreamSynchronize(m->xfer[size_t(i)]));
        }
    }
    RTKM_HIP(hipSetDevice(m->devices[0]));
    rc = grow(&m->gathered[slot], &m->gathered_bytes[slot], part * size_t(n));
    if (rc != RTK_OK) return rc;
    for (int i = 1; i < n; i++) {
        RTKM_HIP(hipSetDevice(m->devices[size_t(i)]));
        rc = grow(&m->compact[slot][size_t(i)], &m->compact_bytes[slot][size_t(i)], part);
        if (rc != RTK_OK) return rc;
    }
    // 1. every device renders its tiles on its render stream -- device 0 straight into its part of the gather target (for
    //    ncclGather: the in-place send buffer of the root) -- once the frame that used this buffer set has been un-permuted
    char* const gathered = static_cast<char*>(m->gathered[slot]);
    for (int i = 0; i < n; i++) {
        RTKM_HIP(hipSetDevice(m->devices[size_t(i)]));
        if (plan[1] && m->slot_used[slot]) RTKM_HIP(hipStreamWaitEvent(m->streams[size_t(i)], m->released[slot], 0));
        rtk_render_opts o = *opts;
        o.rank = i;
        o.n_ranks = n;
        o.stream = m->streams[size_t(i)];
        if (n == 1) o.variant |= 1 << 22;  // one device through the gather path (RCCL forced): still a tile buffer
        void* target = i == 0 ? static_cast<void*>(gathered) : m->compact[slot][size_t(i)];
        rc = rtk_render_device(m->ctxs[size_t(i)], cam, &o, target, nullptr, nullptr);
        if (rc != RTK_OK) return rc;
        RTKM_HIP(hipEventRecord(m->rendered[slot][size_t(i)], m->streams[size_t(i)]));
        RTKM_HIP(hipStreamWaitEvent(m->xfer[size_t(i)], m->rendered[slot][size_t(i)], 0));
    }
    // 2. the one gather, on the transfer streams: the render streams are free for the next frame meanwhile
    if (!m->comms.empty()) {
        const ncclDataType_t type = opts->real_mode == RTK_REAL_F64 ? ncclDouble : ncclFloat;
        const size_t count = size_t(tpr) * 3 * RTK_TILE_PIXELS;
        ncclResult_t r = g_rccl.GroupStart();
        hipError_t he = hipSuccess;
        for (int i = 0; i < n && r == ncclSuccess && he == hipSuccess; i++) {
            he = hipSetDevice(m->devices[size_t(i)]);
            if (he != hipSuccess) break;
            const void* send = i == 0 ? static_cast<const void*>(gathered) : m->compact[slot][size_t(i)];  // root: in place (its part is part 0)
            r = g_rccl.Gather(send, gathered, count, type, 0, m->comms[size_t(i)], m->xfer[size_t(i)]);
        }
        const ncclResult_t r2 = g_rccl.GroupEnd();  // always: an open group would defer every later RCCL call of this thread
        if (he != hipSuccess || r != ncclSuccess || r2 != ncclSuccess) {
            for (int i = 0; i < n; i++) {  // what was launched is allowed to finish before the error is reported
                (void)hipSetDevice(m->devices[size_t(i)]);
                (void)hipStreamSynchronize(m->streams[size_t(i)]);
            }
            if (he != hipSuccess) return fail(RTK_ERR_HIP, "hipSetDevice failed inside the gather group: %s", hipGetErrorString(he));
            return fail(RTK_ERR_HIP, "ncclGather failed: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : r2));
        }
    } else {
        for (int i = 1; i < n; i++) {
            RTKM_HIP(hipSetDevice(m->devices[size_t(i)]));
            RTKM_HIP(hipMemcpyPeerAsync(gathered + part * size_t(i), m->devices[0], m->compact[slot][size_t(i)], m->devices[size_t(i)], part, m->xfer[size_t(i)]));
            RTKM_HIP(hipEventRecord(m->arrived[slot][size_t(i)], m->xfer[size_t(i)]));
        }
        RTKM_HIP(hipSetDevice(m->devices[0]));
        for (int i = 1; i < n; i++) RTKM_HIP(hipStreamWaitEvent(m->xfer[0], m->arrived[slot][size_t(i)], 0));
    }
    // 3. compact tiles -> row-major image + bytes, on the first device's transfer stream (with RCCL the gather itself ordered
    //    it; with peer copies the events above did), then the buffer set is released
    RTKM_HIP(hipSetDevice(m->devices[0]));
    rc = rtk_tiles_unpermute(m->ctxs[0], cam->image_width, cam->image_height, n, opts->real_mode, gathered, d_linear, d_rgb8, m->xfer[0]);
    if (rc != RTK_OK) return rc;
    RTKM_HIP(hipEventRecord(m->released[slot], m->xfer[0]));
    m->slot_used[slot] = true;
    m->frames_enqueued++;
    return RTK_OK;
}

int rtk_multi_wait(rtk_multi* m) {
    if (!m) return fail(RTK_ERR_INVALID, "rtk_multi_wait: null argument");
    const int n = int(m->ctxs.size());
    int rc = rtk::wait_with_progress(m->ctxs.data(), m->streams.data(), n);  // the renders (progress reports come from their counters)
    for (int i = 0; i < n && rc == RTK_OK; i++) {                              // ... then the gathers and the un-permutes
        hipError_t e = hipSetDevice(m->devices[size_t(i)]);
        if (e == hipSuccess) e = hipStreamSynchronize(m->xfer[size_t(i)]);
        if (e != hipSuccess) rc = fail(RTK_ERR_HIP, "rtk_multi_wait: hipStreamSynchronize failed: %s", hipGetErrorString(e));
    }
    (void)hipSetDevice(m->devices[0]);
    return rc;
}

int rtk_render_multi_device(rtk_multi* m, const rtk_camera* cam, const rtk_render_opts* opts, void* d_linear, uint8_t* d_rgb8) {
    if (!m || !cam || !opts) return fail(RTK_ERR_INVALID, "rtk_render_multi_device: null argument");
    const int rc = rtk_render_multi_enqueue(m, cam, opts, d_linear, d_rgb8);
    return rc != RTK_OK ? rc : rtk_multi_wait(m);
}

int rtk_render_multi(rtk_multi* m, const rtk_camera* cam, const rtk_render_opts* opts, double* h_linear, uint8_t* h_rgb8) {
    if (!m || !cam || !opts) return fail(RTK_ERR_INVALID, "rtk_render_multi: null argument");
    if (cam->image_width <= 0 || cam->image_height <= 0) return fail(RTK_ERR_INVALID, "rtk_render_multi: bad image size");
    const size_t n = size_t(cam->image_width) * cam->image_height * 3;
    const size_t elem = opts->real_mode == RTK_REAL_F64 ? sizeof(double) : sizeof(float);
    RTKM_HIP(hipSetDevice(m->devices[0]));
    int rc = grow(&m->image, &m->image_bytes, n * elem);
    if (rc == RTK_OK) rc = grow(reinterpret_cast<void**>(&m->rgb8), &m->rgb8_bytes, n);
    if (rc != RTK_OK) return rc;
    rc = rtk_render_multi_device(m, cam, opts, m->image, m->rgb8);
    if (rc != RTK_OK) return rc;
    RTKM_HIP(hipSetDevice(m->devices[0]));
    if (h_linear) {
        if (opts->real_mode == RTK_REAL_F64) {
            RTKM_HIP(hipMemcpy(h_linear, m->image, n * sizeof(double), hipMemcpyDeviceToHost));
        } else {
            std::vector<float> tmp(n);
            RTKM_HIP(hipMemcpy(tmp.data(), m->image, n * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t k = 0; k < n; k++) h_linear[k] = double(tmp[k]);
        }
    }
    if (h_rgb8) RTKM_HIP(hipMemcpy(h_rgb8, m->rgb8, n, hipMemcpyDeviceToHost));
    return RTK_OK;
}

}  // extern "C"
