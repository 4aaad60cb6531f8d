// rtk_microbench.hip -- measured ceilings of the machine the render kernel runs on, for bench.py's roofline object
// (librtk_microbench.so; a measurement tool, not part of the render path -- librtk_hip.so does not link it).
//
// SURVEY.md 8(d) asks for the achievable HBM rate from a stream copy next to the spec figure; the render kernel's traversal
// program lives in LDS, so the LDS read rate is the ceiling its byte model has to be quoted against; and the VALU-issue
// roofline prices instruction classes by their issue cost, which is measured here instead of assumed:
//
//   hbm_copy_GBps          grid-stride 16-byte copy of a 1 GiB buffer (read + written bytes / time)
//   hbm_read_GBps          the same buffer summed (read only)
//   lds_read_b128_GBps     ds_read_b128, every lane its own consecutive 16 bytes (conflict-free), all CUs
//   lds_read_b128_random_GBps   ... at per-lane pseudo-random 32-byte records (what the box step does)
//   lds_roundtrip_cycles   dependent ds_read_b32 chain: shader cycles per round trip, one wave per SIMD
//   issue_<class>          wave64 instructions issued per shader cycle per SIMD (AGGREGATE: all instructions of the launch / SIMDs /
//                          its duration in cycles of the in-kernel clock), 8 independent chains per lane, at the number of waves
//                          per SIMD (1, 2, 4, 8: issue_v_fma_f32_<n>wave[s]) where v_fma_f32 issues fastest, for v_fma_f32, v_fma_f64, v_mul_f64, v_add_f64, v_mul_lo_u32, v_rcp_f32,
//                          v_sqrt_f32, v_rcp_f64, v_rsq_f64, v_sqrt_f64 (the TRANS classes), v_cndmask_b32, v_max3_f32,
//                          v_pk_fma_f32 / v_pk_mul_f32 (two f32 operations per lane and instruction), v_fmac_f32 / v_mul_f32 (the
//                          two-source VOP2 encodings, against the three-source VOP3 v_fma_f32)
//   shader_clock_GHz       the in-kernel clock (delta s_memtime / delta s_memrealtime x 100 MHz) of the first v_fma_f32 launches (cold chip)
//   clock_dense_valu_GHz   the in-kernel clock of the same kernel after ~0.3 s of back-to-back launches: delta s_memtime /
//                          delta s_memrealtime x 100 MHz (MI355X_MICROARCH.md, DVFS item 6), median over workgroups -- what the
//                          chip sustains with every SIMD issuing v_fma_f32 from 8 waves
//   clock_light_load_GHz   the same quotient in the dependent LDS chain kernel (one wave per CU: a nearly idle chip)
//
// C ABI: rtk_microbench_run(device, out, n) fills out[0..n) in the order of rtk_microbench_names() (comma separated).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace {

__global__ __launch_bounds__(256) void copy_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_kernel(const uint4* __restrict__ src, size_t n, unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (size_t i = size_t(blockIdx.x) * 256 + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345u) *sink = acc;
}

// LDS: 64 KB staged, every lane reads 16 bytes per step; RANDOM: a different 32-byte record each step (LCG), two reads per step
template <bool RANDOM>
__global__ __launch_bounds__(1024) void lds_read_kernel(int iters, unsigned* __restrict__ sink) {
    extern __shared__ __align__(16) unsigned char lds[];
    uint4* words = reinterpret_cast<uint4*>(lds);
    for (int k = threadIdx.x; k < 4096; k += blockDim.x) words[k] = make_uint4(k, k + 1, k + 2, k + 3);
    __syncthreads();
    unsigned acc = 0, at = threadIdx.x * 2654435761u;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            unsigned idx;
            if (RANDOM) {
                at = at * 1664525u + 1013904223u;
                idx = (at >> 20) & 2047u;  // 2048 records of 32 bytes
                const uint4 a = words[2 * idx], b = words[2 * idx + 1];
                acc ^= a.x ^ a.w ^ b.y ^ b.z;
            } else {
                idx = (threadIdx.x + 64u * unsigned(u) + unsigned(it)) & 4095u;
                const uint4 a = words[idx];
                acc ^= a.x ^ a.y ^ a.z ^ a.w;
            }
        }
    }
    if (acc == 0x12345u) *sink = acc;
}

// dependent LDS chain: lds[i] holds the next index
__global__ __launch_bounds__(64) void lds_chain_kernel(int iters, unsigned long long* __restrict__ cycles, unsigned* __restrict__ sink,
                                                       unsigned long long* __restrict__ real_ticks = nullptr) {
    __shared__ unsigned next[4096];
    for (int k = threadIdx.x; k < 4096; k += 64) next[k] = (k * 1031u + 17u) & 4095u;
    __syncthreads();
    unsigned at = threadIdx.x;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) at = next[at];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (threadIdx.x == 0 && real_ticks) real_ticks[blockIdx.x] = r1 - r0;
    if (at == 0xFFFFFFFFu) *sink = at;
}

enum IssueClass { I_FMA_F32, I_FMA_F64, I_MUL_F64, I_ADD_F64, I_MUL_LO_U32, I_RCP_F32, I_SQRT_F32, I_RCP_F64, I_RSQ_F64, I_SQRT_F64, I_CNDMASK, I_MAX3_F32, I_PK_FMA_F32, I_PK_MUL_F32, I_FMAC_F32, I_MUL_F32, I_COUNT };

// 8 independent chains per lane x 8 * kIssueReps instructions per trip (128: the loop's own scalar instructions and its taken
// branch are 2 % of the stream)
constexpr int kIssueReps = 16;
template <int CLS>
__global__ __launch_bounds__(1024) void issue_kernel(int iters, unsigned long long* __restrict__ cycles, float* __restrict__ sink,
                                                    unsigned long long* __restrict__ real_ticks = nullptr) {
    float f[8];
    double d[8];
    unsigned u[8];
    for (int k = 0; k < 8; k++) {
        f[k] = 1.0f + 0.001f * float(threadIdx.x + k);
        d[k] = 1.0 + 0.001 * double(threadIdx.x + k);
        u[k] = threadIdx.x * 3u + unsigned(k) + 1u;
    }
    const float fb = 0.999f, fc = 0.001f;
    const double db = 0.999, dc = 0.001;
    const unsigned um = 0x9E3779B1u;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < kIssueReps; rep++) {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                if (CLS == I_FMA_F32) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fb), "v"(fc));
                if (CLS == I_FMA_F64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k]) : "v"(db), "v"(dc));
                if (CLS == I_MUL_F64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[k]) : "v"(db));
                if (CLS == I_ADD_F64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[k]) : "v"(dc));
                if (CLS == I_MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(um));
                if (CLS == I_RCP_F32) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[k]));
                if (CLS == I_SQRT_F32) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[k]));
                if (CLS == I_RCP_F64) asm volatile("v_rcp_f64 %0, %0" : "+v"(d[k]));
                if (CLS == I_RSQ_F64) asm volatile("v_rsq_f64 %0, %0" : "+v"(d[k]));
                if (CLS == I_SQRT_F64) asm volatile("v_sqrt_f64 %0, %0" : "+v"(d[k]));
                if (CLS == I_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[k]) : "v"(fb));
                if (CLS == I_MAX3_F32) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(fb), "v"(fc));
                if (CLS == I_PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d[k]) : "v"(db), "v"(dc));   // (two f32 lanes in a 64-bit pair)
                if (CLS == I_PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(d[k]) : "v"(db));
                if (CLS == I_FMAC_F32) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(f[k]) : "v"(fb), "v"(fc));   // VOP2: two sources + the accumulator
                if (CLS == I_MUL_F32) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(fb));            // VOP2: two sources
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (threadIdx.x == 0 && real_ticks) real_ticks[blockIdx.x] = r1 - r0;
    float acc = 0.0f;
    for (int k = 0; k < 8; k++) acc += f[k] + float(d[k]) + float(u[k]);
    if (acc == 12345.678f) *sink = acc;
}

const char* kNames =
    "hbm_copy_GBps,hbm_read_GBps,lds_read_b128_GBps,lds_read_b128_random_GBps,lds_roundtrip_cycles,shader_clock_GHz,"
    "issue_v_fma_f32,issue_v_fma_f64,issue_v_mul_f64,issue_v_add_f64,issue_v_mul_lo_u32,issue_v_rcp_f32,issue_v_sqrt_f32,issue_v_rcp_f64,issue_v_rsq_f64,"
    "issue_v_sqrt_f64,issue_v_cndmask_b32,issue_v_max3_f32,issue_v_pk_fma_f32,issue_v_pk_mul_f32,issue_v_fmac_f32,issue_v_mul_f32,clock_dense_valu_GHz,clock_light_load_GHz,"
    "issue_v_fma_f32_1wave,issue_v_fma_f32_2waves,issue_v_fma_f32_4waves,issue_v_fma_f32_8waves";
constexpr int kResults = 6 + I_COUNT + 2 + 4;

#define MB_HIP(call)                                  \
    do {                                              \
        const hipError_t e_ = (call);                 \
        if (e_ != hipSuccess) {                       \
            std::snprintf(g_err, sizeof g_err, "%s: %s", #call, hipGetErrorString(e_)); \
            return -3;                                \
        }                                             \
    } while (0)
char g_err[256] = "";

template <typename F>
int time_ms(F&& launch, int reps, float* best_ms) {
    hipEvent_t a, b;
    MB_HIP(hipEventCreate(&a));
    MB_HIP(hipEventCreate(&b));
    *best_ms = 1e30f;
    for (int r = 0; r < reps; r++) {
        MB_HIP(hipEventRecord(a, nullptr));
        launch();
        MB_HIP(hipEventRecord(b, nullptr));
        MB_HIP(hipEventSynchronize(b));
        float ms = 0;
        MB_HIP(hipEventElapsedTime(&ms, a, b));
        if (ms < *best_ms) *best_ms = ms;
    }
    MB_HIP(hipGetLastError());
    (void)hipEventDestroy(a);
    (void)hipEventDestroy(b);
    return 0;
}

// delta s_memtime / delta s_memrealtime x 100 MHz, median over the workgroups
int clock_from(const unsigned long long* d_cycles, const unsigned long long* d_real, int blocks, double* ghz) {
    std::vector<unsigned long long> cyc(static_cast<size_t>(blocks), 0ull), real(static_cast<size_t>(blocks), 0ull);
    MB_HIP(hipMemcpy(cyc.data(), d_cycles, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    MB_HIP(hipMemcpy(real.data(), d_real, real.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::vector<double> q;
    for (int b = 0; b < blocks; b++)
        if (real[size_t(b)] > 0) q.push_back(double(cyc[size_t(b)]) / double(real[size_t(b)]) * 0.1);
    if (q.empty()) {
        *ghz = 0;
        return 0;
    }
    std::nth_element(q.begin(), q.begin() + q.size() / 2, q.end());
    *ghz = q[q.size() / 2];
    return 0;
}

template <int CLS>
int run_issue(int cus, int wgs_per_cu, int threads, unsigned long long* d_cycles, unsigned long long* d_real, float* d_sink, double* rate, double* clock_ghz) {
    // AGGREGATE rate: every wave instruction of the launch / (SIMDs x the launch's duration in shader cycles), the duration from the
    // launch's event time and the in-kernel clock of the same launch (delta s_memtime / delta s_memrealtime) -- no assumption
    // about how many of the workgroups were resident at once
    const int blocks = cus * wgs_per_cu, iters = 40000 / kIssueReps;
    float ms = 0;
    const int rc = time_ms([&] { issue_kernel<CLS><<<dim3(blocks), dim3(threads), 0, nullptr>>>(iters, d_cycles, d_sink, d_real); }, 3, &ms);
    if (rc != 0) return rc;
    double ghz = 0;
    if (clock_from(d_cycles, d_real, blocks, &ghz) != 0) return -3;
    const double per_simd = double(blocks) * double(threads / 64) * double(iters) * 8.0 * kIssueReps / (double(cus) * 4.0);
    *rate = ghz > 0 ? per_simd / (double(ms) * 1e6 * ghz) : 0.0;
    if (clock_ghz) *clock_ghz = ghz;
    return 0;
}

}  // namespace

extern "C" {

const char* rtk_microbench_names(void) { return kNames; }
const char* rtk_microbench_last_error(void) { return g_err; }
int rtk_microbench_count(void) { return kResults; }

int rtk_microbench_run(int device, double* out, int n_out) {
    if (!out || n_out < kResults) {
        std::snprintf(g_err, sizeof g_err, "rtk_microbench_run: need room for %d results", kResults);
        return -1;
    }
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
        std::snprintf(g_err, sizeof g_err, "no HIP device");
        return -2;
    }
    MB_HIP(hipSetDevice(device));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    for (int k = 0; k < n_out; k++) out[k] = 0.0;
    // ---- HBM
    const size_t bytes = size_t(1) << 30, n16 = bytes / 16;
    uint4 *src = nullptr, *dst = nullptr;
    unsigned* sink = nullptr;
    unsigned long long* d_cycles = nullptr;
    float* d_fsink = nullptr;
    MB_HIP(hipMalloc(reinterpret_cast<void**>(&src), bytes));
    MB_HIP(hipMalloc(reinterpret_cast<void**>(&dst), bytes));
    MB_HIP(hipMalloc(reinterpret_cast<void**>(&sink), 64));
    MB_HIP(hipMalloc(reinterpret_cast<void**>(&d_cycles), size_t(cus) * 8 * sizeof(unsigned long long)));
    MB_HIP(hipMalloc(reinterpret_cast<void**>(&d_fsink), 64));
    MB_HIP(hipMemset(src, 1, bytes));
    MB_HIP(hipMemset(dst, 0, bytes));
    float ms = 0;
    int rc = time_ms([&] { copy_kernel<<<dim3(cus * 16), dim3(256), 0, nullptr>>>(src, dst, n16); }, 5, &ms);
    if (rc == 0) out[0] = 2.0 * double(bytes) / (double(ms) * 1e-3) / 1e9;
    if (rc == 0) rc = time_ms([&] { read_kernel<<<dim3(cus * 16), dim3(256), 0, nullptr>>>(src, n16, sink); }, 5, &ms);
    if (rc == 0) out[1] = double(bytes) / (double(ms) * 1e-3) / 1e9;
    // ---- LDS
    const int lds_iters = 4000;
    if (rc == 0) rc = time_ms([&] { lds_read_kernel<false><<<dim3(cus), dim3(1024), 65536, nullptr>>>(lds_iters, sink); }, 3, &ms);
    if (rc == 0) out[2] = double(cus) * 1024.0 * double(lds_iters) * 8.0 * 16.0 / (double(ms) * 1e-3) / 1e9;
    if (rc == 0) rc = time_ms([&] { lds_read_kernel<true><<<dim3(cus), dim3(1024), 65536, nullptr>>>(lds_iters, sink); }, 3, &ms);
    if (rc == 0) out[3] = double(cus) * 1024.0 * double(lds_iters) * 8.0 * 32.0 / (double(ms) * 1e-3) / 1e9;
    if (rc == 0) {
        const int chain_iters = 20000;
        rc = time_ms([&] { lds_chain_kernel<<<dim3(cus), dim3(64), 0, nullptr>>>(chain_iters, d_cycles, sink); }, 2, &ms);
        if (rc == 0) {
            std::vector<unsigned long long> cyc(static_cast<size_t>(cus), 0ull);
            MB_HIP(hipMemcpy(cyc.data(), d_cycles, cyc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            double mean = 0;
            for (unsigned long long c : cyc) mean += double(c);
            out[4] = mean / double(cus) / double(chain_iters);
        }
    }
    // ---- issue rates
    // v_fma_f32 at 1, 2, 4 (one workgroup per CU) and 8 (two workgroups of 16 waves) waves per SIMD; every class then at the best of them
    unsigned long long* d_real = nullptr;
    MB_HIP(hipMalloc(reinterpret_cast<void**>(&d_real), size_t(cus) * 8 * sizeof(unsigned long long)));
    double clock = 0;
    const int cfg_wgs[4] = {1, 1, 1, 2}, cfg_threads[4] = {256, 512, 1024, 1024};
    int best = 2;
    for (int c = 0; c < 4 && rc == 0; c++) {
        rc = run_issue<I_FMA_F32>(cus, cfg_wgs[c], cfg_threads[c], d_cycles, d_real, d_fsink, &out[6 + I_COUNT + 2 + c], c == 2 ? &clock : nullptr);
        if (out[6 + I_COUNT + 2 + c] > out[6 + I_COUNT + 2 + best]) best = c;
    }
    out[5] = clock;
    const int bw = cfg_wgs[best], bt = cfg_threads[best];
    out[6 + I_FMA_F32] = out[6 + I_COUNT + 2 + best];
    if (rc == 0) rc = run_issue<I_FMA_F64>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_FMA_F64], nullptr);
    if (rc == 0) rc = run_issue<I_MUL_F64>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_MUL_F64], nullptr);
    if (rc == 0) rc = run_issue<I_ADD_F64>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_ADD_F64], nullptr);
    if (rc == 0) rc = run_issue<I_MUL_LO_U32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_MUL_LO_U32], nullptr);
    if (rc == 0) rc = run_issue<I_RCP_F32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_RCP_F32], nullptr);
    if (rc == 0) rc = run_issue<I_SQRT_F32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_SQRT_F32], nullptr);
    if (rc == 0) rc = run_issue<I_RCP_F64>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_RCP_F64], nullptr);
    if (rc == 0) rc = run_issue<I_RSQ_F64>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_RSQ_F64], nullptr);
    if (rc == 0) rc = run_issue<I_SQRT_F64>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_SQRT_F64], nullptr);
    if (rc == 0) rc = run_issue<I_CNDMASK>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_CNDMASK], nullptr);
    if (rc == 0) rc = run_issue<I_MAX3_F32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_MAX3_F32], nullptr);
    if (rc == 0) rc = run_issue<I_PK_FMA_F32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_PK_FMA_F32], nullptr);
    if (rc == 0) rc = run_issue<I_PK_MUL_F32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_PK_MUL_F32], nullptr);
    if (rc == 0) rc = run_issue<I_FMAC_F32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_FMAC_F32], nullptr);
    if (rc == 0) rc = run_issue<I_MUL_F32>(cus, bw, bt, d_cycles, d_real, d_fsink, &out[6 + I_MUL_F32], nullptr);
    // ---- the clock the chip holds (DVFS): dense VALU issue after a warm-up of back-to-back launches, then a nearly idle chip
    if (rc == 0) {
        const int blocks = cus * bw, iters = 40000 / kIssueReps;
        for (int k = 0; k < 100; k++) issue_kernel<I_FMA_F32><<<dim3(blocks), dim3(bt), 0, nullptr>>>(iters, d_cycles, d_fsink, d_real);  // ~3 ms each
        MB_HIP(hipDeviceSynchronize());
        rc = clock_from(d_cycles, d_real, blocks, &out[6 + I_COUNT]);
        if (rc == 0) {
            lds_chain_kernel<<<dim3(cus), dim3(64), 0, nullptr>>>(200000, d_cycles, sink, d_real);
            MB_HIP(hipDeviceSynchronize());
            rc = clock_from(d_cycles, d_real, cus, &out[6 + I_COUNT + 1]);
        }
    }
    (void)hipFree(d_real);
    (void)hipFree(src);
    (void)hipFree(dst);
    (void)hipFree(sink);
    (void)hipFree(d_cycles);
    (void)hipFree(d_fsink);
    return rc;
}

}  // extern "C"
