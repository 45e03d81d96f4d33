// engine.hip -- libacx.so: contexts, device-resident constraint systems, kernel orchestration
// and the C ABI of include/acx.h.  One acx_ctx = one GPU.  Several GPUs: either one process per GPU with the
// collectives in the host layer (acx_ntt_dist_step_dev + RCCL: parallel.py, examples/), or ONE process and
// an acx_mgpu handle that shards the rows, issues the RCCL collectives itself and keeps the reference's
// one-call shape (mgpu.inc.h, included at the end of this file).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <atomic>
#include <list>
#include <array>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/acx.h"
#include "circuit_host.h"
#include "host_field.h"
#include "kernels.hip.h"

namespace acx {
bool launch_ntt_r4(bool bls12_381, int lp, int lg, unsigned tiles, hipStream_t st, const NttPass& Q);      // ntt_r4.hip
}

using namespace acx;

#include "abi_common.inc.h"


// roctx ranges around the blocking ABI calls (SURVEY.md section 5 "tracing"): with ACX_ROCTX=1 every entry point that
// enqueues device work pushes a range named after itself, so a `rocprofv3 --marker-trace --kernel-trace` timeline shows which
// call each kernel belongs to.  Bound by dlopen on first use (librocprofiler-sdk-roctx / libroctx64); off by default: one
// relaxed load per call.
struct AbiRange {
    using PushFn = int (*)(const char*);
    using PopFn = int (*)();
    static void bind(PushFn& push, PopFn& pop) {
        static PushFn p_push = nullptr;
        static PopFn p_pop = nullptr;
        static std::once_flag once;
        std::call_once(once, [] {
            const char* e = std::getenv("ACX_ROCTX");
            if (!e || std::atoi(e) == 0) return;
            for (const char* nm : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
                if (void* so = dlopen(nm, RTLD_NOW | RTLD_GLOBAL)) {
                    p_push = reinterpret_cast<PushFn>(dlsym(so, "roctxRangePushA"));
                    p_pop = reinterpret_cast<PopFn>(dlsym(so, "roctxRangePop"));
                    if (p_push && p_pop) return;
                    p_push = nullptr; p_pop = nullptr;
                }
            }
        });
        push = p_push; pop = p_pop;
    }
    PopFn pop = nullptr;
    explicit AbiRange(const char* name) {
        PushFn push = nullptr;
        bind(push, pop);
        if (push) push(name); else pop = nullptr;
    }
    ~AbiRange() { if (pop) pop(); }
    AbiRange(const AbiRange&) = delete;
    AbiRange& operator=(const AbiRange&) = delete;
};
#define ACX_RANGE() AbiRange acx_range_(__func__)

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? ACX_ERR_OOM : ACX_ERR_HIP,                \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                   \
    } while (0)

// ------------------------------------------------------------------------------------ handles
struct NttCfg {
    int impl = 1;            // 0 tile, 1 r4
    uint32_t tile_log = 12;
    uint32_t direct_tw = 20;
    int n_digits = 0;
    uint32_t digits[4] = {0, 0, 0, 0};
};

struct acx_ctx {
    int field = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    HostField hf;
    std::recursive_mutex mu;                               // caches + the device-pointer (single stream) path
    // Host-buffer entry points (acx_r1cs_verify, acx_r1cs_residuals, acx_qap_h, acx_ntt, acx_qap_columns) block on
    // the GPU; concurrent callers -- `safe` foreign calls from several Haskell capabilities -- each take a LANE:
    // its own HIP stream, result slots and scratch arena, so their copies and kernels overlap.
    struct Lane {
        std::mutex mu;
        hipStream_t stream = nullptr;
        unsigned long long* d_result = nullptr;
        uint32_t* d_err = nullptr;
        void* h_slot = nullptr;                // page-locked host copy of the call's result slot (cur_hslot)
        void* arena = nullptr;
        size_t arena_bytes = 0;
        uint4* ntt_scratch = nullptr;
        size_t ntt_scratch_bytes = 0;
        hipStream_t copy_stream = nullptr;     // device-to-host copies that overlap the next batch's kernels
        hipEvent_t ev[2] = {nullptr, nullptr};
        std::vector<void*> pins;               // coset-table entries this lane's current call holds (acx_ctx::CosetTables*)
    };
    static constexpr int kLanes = 4;
    Lane lanes[kLanes];
    std::atomic<unsigned> lane_ticket{0};
    std::map<std::pair<uint32_t, int>, uint4*> twiddles;  // (log_m, inverse) -> omega_M^j, j < M
    std::map<std::pair<uint32_t, int>, uint4*> tw_low;    // (log_n, inverse) -> omega_N^j, j < 1024
    std::map<std::tuple<uint32_t, uint64_t, int, uint32_t>, uint4*> tw_scaled;   // (log_m, count, inverse, log_n of folded 1/N)
    std::map<std::pair<uint32_t, int>, uint4*> tw_limbs;  // (log_m, inverse) -> omega_M^j, j < M/2, limb form (k_ntt_r4)
    // closing-factor tables of the distributed steps in store order (k_dist_table): (log_n, log_r, world, rank, kind, coset base)
    struct DistKey {
        uint32_t log_n, log_r, world, rank; int kind; H256 base;
        bool operator<(const DistKey& o) const {
            return std::tie(log_n, log_r, world, rank, kind, base.l[0], base.l[1], base.l[2], base.l[3]) <
                   std::tie(o.log_n, o.log_r, o.world, o.rank, o.kind, o.base.l[0], o.base.l[1], o.base.l[2], o.base.l[3]);
        }
    };
    std::map<DistKey, uint4*> tw_dist;
    // ... a small LRU as well (kDistCap entries; an h(x) pipeline holds three per size): a caller that varies the coset shift
    // of acx_mgpu_ntt / acx_ntt_dist_step_*_dev must not grow device memory without bound (32 N / world bytes per entry).
    // The tables are only used by launches issued under ctx->mu on ctx->stream, so eviction needs no pins: synchronise, free.
    static constexpr size_t kDistCap = 12;
    std::map<DistKey, uint64_t> tw_dist_stamp;
    std::map<std::pair<uint32_t, std::array<uint64_t, 4>>, uint4*> h_scale;   // (log_n, coset shift) -> {1/z, -1/z} of the h(x) pipeline (get_h_scale)
    NttCfg ntt;
    bool small_coeff = true;                               // use the small-coefficient SELL form where a matrix allows it
    uint4* ntt_scratch = nullptr;                          // ping-pong buffer of the multi-pass NTT
    size_t ntt_scratch_bytes = 0;
    struct CosetTables {                                   // g^j (j < 1024), g^(1024 j) for one (g, log_n, scaled)
        uint4 *lo = nullptr, *hi = nullptr;
        H256 base{{0, 0, 0, 0}};
        uint32_t log_n = 0;
        int scaled = 0;
        int direct = 0;                                    // lo = the full table g^j, j < 2^log_n (hi unused)
        uint64_t stamp = 0;
        int pins = 0;                                      // lanes that hold the pointers (released after their stream drained)
    };
    // A small LRU.  Entries handed to a lane are PINNED until that lane's call has drained its stream (LaneGuard): a lane
    // launches after get_coset_tables has returned and released ctx->mu, so an unpinned entry could be evicted and freed by
    // another lane in between.  Only unpinned entries are evicted; when every entry is pinned the list grows past kCosetCap
    // and shrinks again on later misses.  (The device-pointer path launches under ctx->mu and needs no pin.)
    static constexpr size_t kCosetCap = 8;
    std::list<CosetTables> cosets;
    uint64_t coset_clock = 0;
    unsigned long long* d_result = nullptr;                // {n_bad, first_bad}
    uint32_t* d_err = nullptr;
    void* h_slot = nullptr;                                // page-locked host copy of the result slot, calls without a lane (under mu)
    int n_cu = 256;
};

using CtxLock = std::lock_guard<std::recursive_mutex>;

// The lane the calling thread holds (host-buffer entry points), or null on the device-pointer path.
static thread_local acx_ctx::Lane* t_lane = nullptr;
static inline hipStream_t cur_stream(const acx_ctx* c) { return t_lane ? t_lane->stream : c->stream; }
static inline unsigned long long* cur_result(const acx_ctx* c) { return t_lane ? t_lane->d_result : c->d_result; }
static inline uint32_t* cur_err(const acx_ctx* c) { return t_lane ? t_lane->d_err : c->d_err; }
static inline void* cur_hslot_raw(const acx_ctx* c) { return t_lane ? t_lane->h_slot : c->h_slot; }

struct LaneGuard {
    acx_ctx::Lane* lane = nullptr;
    acx_ctx* ctx = nullptr;
    explicit LaneGuard(acx_ctx* c) : ctx(c) {
        for (int i = 0; i < acx_ctx::kLanes && !lane; ++i)
            if (c->lanes[i].mu.try_lock()) lane = &c->lanes[i];
        if (!lane) {
            lane = &c->lanes[c->lane_ticket.fetch_add(1) % acx_ctx::kLanes];
            lane->mu.lock();
        }
        t_lane = lane;
    }
    ~LaneGuard() {
        if (!lane->pins.empty()) {
            // every successful call has synchronised its stream already; a failed one may still have kernels in flight
            (void)hipStreamSynchronize(lane->stream);
            CtxLock lock(ctx->mu);
            for (void* p : lane->pins) --static_cast<acx_ctx::CosetTables*>(p)->pins;
            lane->pins.clear();
        }
        t_lane = nullptr;
        lane->mu.unlock();
    }
    LaneGuard(const LaneGuard&) = delete;
    LaneGuard& operator=(const LaneGuard&) = delete;
};

struct DevMatrix {
    u32* ptr = nullptr;   // rowptr (CSR) or colptr (CSC)
    u32* idx = nullptr;   // col (CSR) or row (CSC)
    uint4* val = nullptr; // dev format
    u32* colid = nullptr; // CSC only: column of every entry
    uint64_t nnz = 0;
    std::vector<uint32_t> h_ptr;   // CSC only: host copy of colptr (qap_columns_core sorts a batch into sparse and dense columns)
};

constexpr int kRowTiers = 4;

struct acx_r1cs {
    acx_ctx* ctx = nullptr;
    uint64_t n = 0, m = 0;
    uint32_t log_n = 0;
    DevMatrix M[3];
    DevMatrix T[3];        // CSC, built lazily for acx_qap_columns
    bool unit_c = false;   // every stored C value is 1: the kernel never reads C's value stream
    uint32_t small = 0;    // bit k: every coefficient of matrix k's SELL rows is small (|c| <= 2^27): no value stream
    // SELL-64 layout used by the residual kernel (kernels.hip.h)
    u32* sell_ofs[3] = {nullptr, nullptr, nullptr};
    uint2* sell_tail[3] = {nullptr, nullptr, nullptr};
    uint4* sell_val[3] = {nullptr, nullptr, nullptr};
    u32* perm = nullptr;
    u32* long_rows = nullptr;
    uint32_t n_slices = 0, n_long = 0;
    uint32_t tier_rows[4] = {0, 0, 0, 0};               // long_rows by length tier: <= 12, <= 24, <= 48 entries, longer
    // device evaluation plan (present when the system was built from a single-assignment circuit)
    bool has_plan = false;
    const acx_circuit* plan_src = nullptr;           // circuit the plan will be derived from on first acx_r1cs_eval (holds a reference)
    std::vector<uint64_t> plan_order;                // root order the rows were loaded in
    std::vector<uint32_t> plan_level_ofs;
    std::vector<uint8_t> plan_written, plan_kind;   // host copies for argument checks
    std::vector<uint32_t> plan_eq_split_inputs;     // flat input wire of every Equal / Split gate
    uint64_t plan_n_in = 0;
    u32 *ev_items = nullptr, *ev_row = nullptr, *ev_wire_ofs = nullptr, *ev_wires = nullptr;
    uint8_t* ev_kind = nullptr;
    uint4* ev_mul = nullptr;             // per plan item: the Mul gate's record (k_eval_level)
    u32* ev_cols = nullptr;              // per plan item: kEvalLanes columns (k_eval_level_lanes)
    u32* ev_level_ofs = nullptr;         // plan_level_ofs on the device (k_eval_levels_fused)
    u32* ev_equal = nullptr;             // Equal gates whose magic wires k_eval_magic fills after the last level (n_ev_equal of them)
    uint32_t n_ev_equal = 0;
    bool ev_defer_magic = false;
    bool has_csc = false;
    // Device memory of a loaded system in TWO allocations (hipMalloc synchronises the device and costs ~7 us: 22 of them and
    // seven stream waits were most of acx_circuit_to_r1cs on a 2^10-gate circuit): `slab` holds M[k].{ptr, idx, val}, d_w and
    // d_hscale; `sell_slab` holds perm, long_rows and sell_{ofs, tail, val}[k].  The members point into them and are not freed
    // one by one (free_r1cs_device).
    void* slab = nullptr;
    void* sell_slab = nullptr;
    void* csc_slab = nullptr;        // T[k].{ptr, idx, colid, val} of a system whose column views were built on the device (build_csc);
                                     // the column slices of acx_mgpu own their T[k] members one by one (r1cs_column_slice_from_host)
    uint4* d_w = nullptr;  // the witness acx_r1cs_eval leaves resident (m elements); acx_naive_h uses it as scratch
    uint4* d_w_canon = nullptr;                      // conversion target of acx_r1cs_eval's witness download (first use; hipMalloc / hipFree synchronise the device)
    bool resident_valid = false;                     // d_w holds a witness produced by acx_r1cs_eval
    uint4* qh = nullptr;   // h(x) pipeline scratch, 5N elements (allocated on first use)
    uint4* d_hscale = nullptr;       // {1/z, -1/z} as dev elements: the factors the h(x) pipeline lets ride on the stored dot products
};

struct acx_batch {
    acx_ctx* ctx = nullptr;
    std::vector<acx_r1cs*> systems;
    std::vector<const uint4*> witnesses;
    std::vector<ResidualOut> outs;
    SellSystem* d_systems = nullptr;
    uint32_t max_slices = 0;
};

struct acx_naive {          // createPolynomials state for arbitrary distinct roots (n <= 4096)
    acx_r1cs* r = nullptr;
    uint32_t n = 0;
    uint4* roots = nullptr;  // [n] dev
    uint4* tcoef = nullptr;  // [n + 1] target polynomial, dev
    uint4* winv = nullptr;   // [n]
    uint4* Q = nullptr;      // [n][n]
};

namespace {

// Waits for a stream when it goes out of scope.  Declare it AFTER the host objects that enqueued copies read from (locals are
// destroyed in reverse order): then no exit of the function, an error return included, leaves a copy from freed host memory in
// flight.  On the normal path the function has waited already and this is a no-op of ~2 us.
struct StreamDrain {
    hipStream_t s;
    explicit StreamDrain(hipStream_t st) : s(st) {}
    ~StreamDrain() { (void)hipStreamSynchronize(s); }
    StreamDrain(const StreamDrain&) = delete;
    StreamDrain& operator=(const StreamDrain&) = delete;
};

struct DevBuf {  // RAII scratch
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) {
        HIP_TRY(hipMalloc(&p, bytes ? bytes : 16));
        return ACX_OK;
    }
    template <class T> T* as() { return static_cast<T*>(p); }
};

inline int grid_for(const acx_ctx* c, uint64_t work_items, int per_cu = 8) {
    const uint64_t blocks = (work_items + kBlock - 1) / kBlock;
    const uint64_t cap = (uint64_t)c->n_cu * per_cu;
    return (int)std::max<uint64_t>(1, std::min(blocks, cap));
}

inline FeArg dev_arg(const HostField& hf, const H256& mont) {
    FeArg a;
    hf.to_dev_limbs(mont, a.l);
    return a;
}

inline uint32_t ceil_log2(uint64_t n) {
    uint32_t k = 0;
    while ((1ull << k) < n) ++k;
    return k;
}

// ---- field dispatch ---------------------------------------------------------------------
#define DISPATCH_FIELD(ctx, ...)                          \
    do {                                                  \
        if ((ctx)->field == ACX_FIELD_BN254_FR) { using F = Bn254Fr; __VA_ARGS__; }        \
        else { using F = Bls12381Fr; __VA_ARGS__; }       \
    } while (0)

// Scratch of the calling thread's lane, grown on demand (hipMalloc / hipFree synchronise the whole device, so the
// steady state must not allocate).  One reservation per entry-point call; carve it with the returned base.
int lane_reserve(acx_ctx* c, size_t bytes, uint8_t** base) {
    acx_ctx::Lane* ln = t_lane;
    if (!ln) return fail(ACX_ERR_INVALID_ARG, "internal: no lane");
    if (ln->arena_bytes < bytes) {
        HIP_TRY(hipStreamSynchronize(ln->stream));
        if (ln->arena) (void)hipFree(ln->arena);
        ln->arena = nullptr; ln->arena_bytes = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        HIP_TRY(hipMalloc(&ln->arena, want));
        ln->arena_bytes = want;
    }
    *base = static_cast<uint8_t*>(ln->arena);
    return ACX_OK;
}
inline size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

int launch_convert(acx_ctx* c, bool to_dev, const void* in, void* out, uint64_t count, uint32_t* d_err) {
    if (count == 0) return ACX_OK;
    const int grid = grid_for(c, count);
    DISPATCH_FIELD(c, {
        if (to_dev) hipLaunchKernelGGL((k_convert<F, true>), dim3(grid), dim3(kBlock), 0, cur_stream(c),
                                       (const uint4*)in, (uint4*)out, count, d_err);
        else hipLaunchKernelGGL((k_convert<F, false>), dim3(grid), dim3(kBlock), 0, cur_stream(c),
                                (const uint4*)in, (uint4*)out, count, d_err);
    });
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

// Upload canonical host elements and convert to dev format in place; checks canonicity.
int upload_elements(acx_ctx* c, const acx_fr* host, uint64_t count, uint4* d_out) {
    if (count == 0) return ACX_OK;
    HIP_TRY(hipMemsetAsync(cur_err(c), 0, 4, cur_stream(c)));
    HIP_TRY(hipMemcpyAsync(d_out, host, count * 32, hipMemcpyHostToDevice, cur_stream(c)));
    ACX_TRY(launch_convert(c, true, d_out, d_out, count, cur_err(c)));
    uint32_t err = 0;
    HIP_TRY(hipMemcpyAsync(&err, cur_err(c), 4, hipMemcpyDeviceToHost, cur_stream(c)));
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));
    if (err) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    return ACX_OK;
}

// The same without the host round trip, for entry points that end with a result fetch anyway: the call's slot
// {n_bad, first_bad, canonicity flag} is initialised by ONE 32-byte copy (begin_call), the conversion raises the flag on the
// device, and end_call fetches all three words with ONE copy before the single stream synchronisation -- a small
// system's verify is five enqueues and one wait.
struct CallSlot { unsigned long long n_bad, first_bad; uint32_t noncanonical, pad[3]; };
static_assert(sizeof(CallSlot) == 32, "slot layout");

int begin_call(acx_ctx* c) {
    static const CallSlot init{0ull, ~0ull, 0u, {0u, 0u, 0u}};
    HIP_TRY(hipMemcpyAsync(cur_result(c), &init, sizeof(init), hipMemcpyHostToDevice, cur_stream(c)));
    return ACX_OK;
}
int upload_elements_async(acx_ctx* c, const acx_fr* host, uint64_t count, uint4* d_out) {   // after begin_call
    if (count == 0) return ACX_OK;
    HIP_TRY(hipMemcpyAsync(d_out, host, count * 32, hipMemcpyHostToDevice, cur_stream(c)));
    return launch_convert(c, true, d_out, d_out, count, cur_err(c));
}
// Where a call's result slot lands on the host: page-locked memory of the lane (of the context for calls under ctx->mu).  The
// 32-byte copy back + wait that ends every blocking call takes 16 us into page-locked memory and 26 us into a stack variable
// (tools/microbench/pcie_rates.hip: the runtime stages pageable destinations); the lane / the context lock is held until the
// call has read it.
static inline CallSlot& cur_hslot(const acx_ctx* c) { return *static_cast<CallSlot*>(cur_hslot_raw(c)); }
inline int end_call_fetch(acx_ctx* c, CallSlot* host) {      // the caller synchronises the stream afterwards
    HIP_TRY(hipMemcpyAsync(host, cur_result(c), sizeof(CallSlot), hipMemcpyDeviceToHost, cur_stream(c)));
    return ACX_OK;
}

int download_elements(acx_ctx* c, const uint4* d_in, uint64_t count, acx_fr* host, uint4* d_scratch) {
    if (count == 0) return ACX_OK;
    ACX_TRY(launch_convert(c, false, d_in, d_scratch, count, nullptr));
    HIP_TRY(hipMemcpyAsync(host, d_scratch, count * 32, hipMemcpyDeviceToHost, cur_stream(c)));
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));
    return ACX_OK;
}

// omega_M^j for j < M = 2^log_m (inverse: omega_M^-j), cached.  Caller holds ctx->mu.
int get_pow_table(acx_ctx* c, uint32_t log_m, int inverse, uint4** out) {
    CtxLock lock(c->mu);
    auto key = std::make_pair(log_m, inverse);
    auto it = c->twiddles.find(key);
    if (it != c->twiddles.end()) { *out = it->second; return ACX_OK; }
    const uint64_t count = 1ull << log_m;
    uint4* tw = nullptr;
    HIP_TRY(hipMalloc((void**)&tw, count * 32));
    H256 w = c->hf.root_of_unity((int)log_m);
    if (inverse) w = c->hf.inv(w);
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pow_table<F>), dim3(grid_for(c, count)), dim3(kBlock), 0, cur_stream(c), tw,
                                         count, dev_arg(c->hf, w)));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));   // other lanes may use the table from their own streams
    c->twiddles[key] = tw;
    *out = tw;
    return ACX_OK;
}

// omega_N^j for j < 1024 (low level of the two-level twiddle table), cached.
int get_low_table(acx_ctx* c, uint32_t log_n, int inverse, uint4** out) {
    CtxLock lock(c->mu);
    auto key = std::make_pair(log_n, inverse);
    auto it = c->tw_low.find(key);
    if (it != c->tw_low.end()) { *out = it->second; return ACX_OK; }
    uint4* tw = nullptr;
    HIP_TRY(hipMalloc((void**)&tw, 1024 * 32));
    H256 w = c->hf.root_of_unity((int)log_n);
    if (inverse) w = c->hf.inv(w);
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pow_table<F>), dim3(4), dim3(kBlock), 0, cur_stream(c), tw, (u64)1024,
                                         dev_arg(c->hf, w)));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));   // other lanes may use the table from their own streams
    c->tw_low[key] = tw;
    *out = tw;
    return ACX_OK;
}

// omega_M^j for j < M/2 in limb form (sub-transform twiddles of k_ntt_r4), cached.
int get_limb_table(acx_ctx* c, uint32_t log_m, int inverse, uint4** out) {
    CtxLock lock(c->mu);
    auto key = std::make_pair(log_m, inverse);
    auto it = c->tw_limbs.find(key);
    if (it != c->tw_limbs.end()) { *out = it->second; return ACX_OK; }
    const uint64_t count = std::max<uint64_t>(1, (1ull << log_m) / 2);
    uint4* tw = nullptr;
    HIP_TRY(hipMalloc((void**)&tw, count * 16 * kLimbEntryQuads));
    H256 w = c->hf.root_of_unity((int)log_m);
    if (inverse) w = c->hf.inv(w);
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pow_table_limbs<F>), dim3(grid_for(c, count)), dim3(kBlock), 0, cur_stream(c), tw,
                                         count, dev_arg(c->hf, w)));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));   // other lanes may use the table from their own streams
    c->tw_limbs[key] = tw;
    *out = tw;
    return ACX_OK;
}

// first * omega_M^j for j < count (M = 2^log_m; omega^-1 when inverse); first = 1 or 1/2^scaled_log_n.  Cached.
int get_scaled_table(acx_ctx* c, uint32_t log_m, uint64_t count, int inverse, uint32_t scaled_log_n, uint4** out) {
    CtxLock lock(c->mu);
    if (scaled_log_n == 0 && count == (1ull << log_m)) return get_pow_table(c, log_m, inverse, out);
    if (scaled_log_n == 0 && count == 1024) return get_low_table(c, log_m, inverse, out);
    const auto key = std::make_tuple(log_m, count, inverse, scaled_log_n);
    auto it = c->tw_scaled.find(key);
    if (it != c->tw_scaled.end()) { *out = it->second; return ACX_OK; }
    uint4* tw = nullptr;
    HIP_TRY(hipMalloc((void**)&tw, count * 32));
    H256 w = c->hf.root_of_unity((int)log_m);
    if (inverse) w = c->hf.inv(w);
    const H256 first = c->hf.inv(c->hf.from_u64(1ull << scaled_log_n));
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pow_table_scaled<F>), dim3(grid_for(c, count)), dim3(kBlock), 0, cur_stream(c), tw,
                                         count, dev_arg(c->hf, w), dev_arg(c->hf, first)));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));   // other lanes may use the table from their own streams
    c->tw_scaled[key] = tw;
    *out = tw;
    return ACX_OK;
}

// g^j (j < 1024) and g^(1024 j) (j < N/1024) for the coset factor.  A small cache: the h(x) pipeline alternates
// between g (forward) and 1/g with 1/N folded in (inverse) on every call.
// scaled: the low table carries the factor 1/2^log_n (closing multiplication of an inverse coset transform).
// direct: ONE table of all 2^log_n powers (32 bytes each): the closing multiplication then needs no second product.
int get_coset_tables(acx_ctx* c, const H256& base_mont, uint32_t log_n, int scaled, uint4** lo, uint4** hi, int direct = 0) {
    CtxLock lock(c->mu);
    auto hand_out = [&](acx_ctx::CosetTables& e) {
        e.stamp = ++c->coset_clock;
        if (t_lane) { ++e.pins; t_lane->pins.push_back(&e); }
        *lo = e.lo; *hi = e.hi;
        return ACX_OK;
    };
    for (auto& e : c->cosets)
        if (e.base == base_mont && e.log_n == log_n && e.scaled == scaled && e.direct == direct) return hand_out(e);
    // make room: drop least recently used UNPINNED entries while the list is at its cap
    while (c->cosets.size() >= acx_ctx::kCosetCap) {
        auto victim = c->cosets.end();
        for (auto it = c->cosets.begin(); it != c->cosets.end(); ++it)
            if (it->pins == 0 && (victim == c->cosets.end() || it->stamp < victim->stamp)) victim = it;
        if (victim == c->cosets.end()) break;                        // everything is in use: grow
        HIP_TRY(hipDeviceSynchronize());           // kernels already launched with the old tables (device-pointer path, finished lanes)
        (void)hipFree(victim->lo);
        if (victim->hi) (void)hipFree(victim->hi);
        c->cosets.erase(victim);
    }
    // build the tables first; the entry (and its key) exists only once they are complete
    acx_ctx::CosetTables fresh;
    const uint64_t hi_count = (!direct && log_n > 10) ? (1ull << (log_n - 10)) : 0;
    const uint64_t lo_count = direct ? (1ull << log_n) : 1024;
    auto build = [&]() -> int {
        HIP_TRY(hipMalloc((void**)&fresh.lo, lo_count * 32));
        const H256 first = scaled ? c->hf.inv(c->hf.from_u64(1ull << log_n)) : c->hf.one();
        DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pow_table_scaled<F>), dim3(grid_for(c, lo_count)), dim3(kBlock), 0, cur_stream(c), fresh.lo,
                                             lo_count, dev_arg(c->hf, base_mont), dev_arg(c->hf, first)));
        if (hi_count) {
            HIP_TRY(hipMalloc((void**)&fresh.hi, hi_count * 32));
            const H256 b1024 = c->hf.pow_u64(base_mont, 1024);
            DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pow_table<F>), dim3(grid_for(c, hi_count)), dim3(kBlock), 0, cur_stream(c),
                                                 fresh.hi, hi_count, dev_arg(c->hf, b1024)));
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(cur_stream(c)));   // other lanes may use the tables from their own streams
        return ACX_OK;
    };
    const int rc = build();
    if (rc != ACX_OK) {
        (void)hipStreamSynchronize(cur_stream(c));
        if (fresh.lo) (void)hipFree(fresh.lo);
        if (fresh.hi) (void)hipFree(fresh.hi);
        return rc;
    }
    fresh.base = base_mont; fresh.log_n = log_n; fresh.scaled = scaled; fresh.direct = direct;
    c->cosets.push_back(fresh);
    return hand_out(c->cosets.back());
}

inline uint64_t pow2_floor(uint64_t x) { uint64_t p = 1; while (p * 2 <= x) p *= 2; return p; }
inline uint32_t ilog2(uint64_t x) { uint32_t k = 0; while ((1ull << (k + 1)) <= x) ++k; return k; }

// ---- NTT planning ---------------------------------------------------------------------------------
// A length-2^log_n transform is factored into P digits; pass p transforms digit p (tile kernel) and
// multiplies by the inter-pass twiddle.  Two kernel families: k_ntt_tile (<= 8 bits per pass, one
// radix-2 stage per LDS round trip; every size) and k_ntt_r4 (<= 12 bits per pass, four elements
// per lane in registers; log_n >= 10).  Tunables (development / A-B measurements), read once per
// context: ACX_NTT_IMPL=tile|r4, ACX_NTT_TILE_LOG (max log2 elements per r4
// tile, default 12), ACX_NTT_DIRECT_TW (largest log2 size of a direct inter-pass twiddle table,
// default 20), ACX_NTT_DIGITS="10,10" (forces the digit split of every transform of that size).
NttCfg ntt_cfg_from_env() {
    NttCfg g;
    if (const char* e = std::getenv("ACX_NTT_IMPL")) g.impl = std::string(e) == "tile" ? 0 : 1;
    if (const char* e = std::getenv("ACX_NTT_TILE_LOG")) g.tile_log = (uint32_t)std::max(6, std::min(12, std::atoi(e)));
    if (const char* e = std::getenv("ACX_NTT_DIRECT_TW")) g.direct_tw = (uint32_t)std::max(0, std::min(24, std::atoi(e)));
    if (const char* e = std::getenv("ACX_NTT_DIGITS")) {
        for (const char* q = e; *q && g.n_digits < 4;) {
            g.digits[g.n_digits++] = (uint32_t)std::strtoul(q, const_cast<char**>(&q), 10);
            if (*q == ',') ++q;
        }
    }
    return g;
}

// (LP, LG) instances of k_ntt_r4 that are compiled (ntt_r4.hip)
inline int r4_pick_lg(int lp, int want) {      // largest compiled LG <= want, or -1
    static const int kLg[4][3] = {{0, 2, 4}, {0, 2, -1}, {0, 1, 2}, {0, -1, -1}};
    const int* row = kLg[(lp - 6) / 2];
    int best = -1;
    for (int i = 0; i < 3; ++i) if (row[i] >= 0 && row[i] <= want) best = std::max(best, row[i]);
    return best;
}

// In-place batched NTT on dev-format data.  Caller holds ctx->mu.
//   forward: X[k] = sum_i x[i] (shift * omega^k)^i      inverse: undoes it.
// post_mont (inverse transforms without a coset shift only): the coefficients are multiplied by post^i on the way out --
// "interpolate, then move to the coset post*<omega>" in one closing multiplication (the h(x) pipeline).  Returns
// ACX_ERR_UNSUPPORTED when this size has no such fused form; the caller then takes the two-step route.
// post_batches (with post_mont): only the first post_batches vectors of the batch take the post factor, the others end as a
// plain inverse transform; *post_limited reports whether this plan could do that (it needs 1/N folded into the twiddles, i.e.
// two or more passes) -- if not, every vector takes the factor.
// in_a, in_b (both or neither; batch 1, two or more r4 passes): the transform of the POINTWISE PRODUCT in_a[i] * in_b[i] lands in d
// -- the product is formed as the first pass loads its points, no vector of products ever exists.  add_out: a vector added to
// the output behind the closing step, d[k] = X[k] + add_out[k].  (h(x): the last transform takes L * R on the way in and the
// coefficient-domain -O/z on the way out.)  ACX_ERR_UNSUPPORTED when the plan of this size cannot do it: nothing was launched.
int ntt_dev_locked(acx_ctx* c, uint4* d, uint32_t log_n, uint64_t batch, int inverse, const H256* shift_mont,
                   const H256* post_mont = nullptr, uint64_t post_batches = 0, bool* post_limited = nullptr,
                   const uint4* in_a = nullptr, const uint4* in_b = nullptr, const uint4* add_out = nullptr) {
    if (post_limited) *post_limited = false;
    if ((int)log_n > c->hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "log_n exceeds the field's two-adicity");
    if (batch == 0) return ACX_OK;
    const HostField& hf = c->hf;
    const NttCfg& cfg = c->ntt;
    const uint64_t N = 1ull << log_n;
    const uint64_t batch_pow2 = batch & (~batch + 1);   // largest power of two dividing batch
    // ---- digits
    bool r4 = cfg.impl == 1 && log_n >= 10 && log_n <= 36;
    int P = 0;
    uint32_t lg[4] = {0, 0, 0, 0};
    if (r4) {
        uint32_t sum = 0;
        for (int i = 0; i < cfg.n_digits; ++i) sum += cfg.digits[i];
        if (cfg.n_digits && sum == log_n) {
            P = cfg.n_digits;
            for (int i = 0; i < P; ++i) lg[i] = cfg.digits[i];
        } else if (log_n <= 12 && (log_n % 2 == 0 || batch_pow2 >= 2) && (log_n <= 10 || batch >= 128)) {
            P = 1; lg[0] = log_n;                         // one workgroup per transform: right once a batch fills the chip
        } else if (log_n <= 12) {
            // few transforms of 2^11 / 2^12 points: a single 1024-thread workgroup per transform leaves the chip idle
            // (2^12: 55 us alone, 19 us per transform in a batch of 3); two passes spread the work (25 us, 8.8 us).
            // Also the odd single transform, whose 5-bit pass brings the column pairs.
            P = 2; lg[0] = log_n - 5; lg[1] = 5;
        } else if (log_n <= 17) {
            P = 2; lg[0] = log_n - 8; lg[1] = 8;          // measured best (tools/ntt_sweep.sh, profiles/r02_ntt_plans.txt)
        } else if (log_n <= 20) {
            P = 2; lg[0] = log_n % 2 ? 7 : 8; lg[1] = log_n - lg[0];
        } else if (log_n <= 28) {
            P = 3; lg[0] = log_n - 16; lg[1] = 8; lg[2] = 8;
        } else {
            P = 3;
            for (int p = 0; p < P; ++p) lg[p] = log_n / P + ((uint32_t)p < log_n % P ? 1 : 0);
        }
        for (int p = 0; p < P; ++p) if (lg[p] < 5 || lg[p] > 12) r4 = false;
    }
    if (!r4) {
        P = log_n <= 8 ? 1 : (int)((log_n + 7) / 8);
        for (int p = 0; p < P; ++p) lg[p] = log_n / P + ((uint32_t)p < log_n % P ? 1 : 0);
    }
    uint64_t Wt[4], Vt[4];   // input / output weight of each digit
    for (int p = 0; p < P; ++p) {
        Wt[p] = 1; Vt[p] = 1;
        for (int q = p + 1; q < P; ++q) Wt[p] <<= lg[q];
        for (int q = 0; q < p; ++q) Vt[p] <<= lg[q];
    }
    if ((in_a || in_b || add_out) && (!r4 || P < 2 || batch != 1 || !in_a != !in_b))
        return fail(ACX_ERR_UNSUPPORTED, "no fused product / sum for this transform");
    uint4* scratch = nullptr;
    if (P > 1) {
        const size_t need = (size_t)batch * N * 32;
        uint4*& buf = t_lane ? t_lane->ntt_scratch : c->ntt_scratch;        // ping-pong buffer of this stream
        size_t& have = t_lane ? t_lane->ntt_scratch_bytes : c->ntt_scratch_bytes;
        if (have < need) {
            HIP_TRY(hipStreamSynchronize(cur_stream(c)));
            if (buf) (void)hipFree(buf);
            buf = nullptr; have = 0;
            HIP_TRY(hipMalloc((void**)&buf, need));
            have = need;
        }
        scratch = buf;
    }
    // the r4 kernel can finish with a plain reduction: 1/N of an inverse transform is folded into the last
    // inter-pass twiddle table
    const bool fold_scale = r4 && inverse && !shift_mont && P >= 2;
    // closing coset factor from ONE direct table (one product per element instead of two) where it fits
    const bool direct_coset = r4 && inverse && (shift_mont || post_mont) && log_n <= std::max<uint32_t>(cfg.direct_tw, 16);
    if (post_mont && (!inverse || shift_mont || !direct_coset)) return fail(ACX_ERR_UNSUPPORTED, "no fused post-scale for this transform");
    uint4 *sc_lo = nullptr, *sc_hi = nullptr;
    if (shift_mont) {
        const H256 base = inverse ? hf.inv(*shift_mont) : *shift_mont;
        ACX_TRY(get_coset_tables(c, base, log_n, inverse ? 1 : 0, &sc_lo, &sc_hi, direct_coset ? 1 : 0));
    } else if (post_mont) {
        ACX_TRY(get_coset_tables(c, *post_mont, log_n, fold_scale ? 0 : 1, &sc_lo, &sc_hi, 1));
    }
    for (int p = 0; p < P; ++p) {
        NttPass Q;
        std::memset(&Q, 0, sizeof(Q));
        const bool last = p == P - 1, first = p == 0;
        Q.src = first ? (in_a ? in_a : d) : scratch;
        Q.dst = last ? d : scratch;
        if (first && in_b) Q.mul_src = in_b;
        if (last && add_out) Q.add_src = add_out;
        Q.log_s = lg[p];
        if (lg[p] > 0) {
            uint4* st = nullptr;
            if (r4) ACX_TRY(get_limb_table(c, lg[p], inverse, &st)); else ACX_TRY(get_pow_table(c, lg[p], inverse, &st));
            Q.sub_tw = st;
        }
        Q.idx_mask = N - 1;
        Q.sc_lo = sc_lo; Q.sc_hi = sc_hi;
        const uint64_t S = 1ull << lg[p];
        // columns a tile may take (powers of two), and the tile's element budget
        const uint64_t col_avail = P == 1 ? batch_pow2 : (!last ? (1ull << lg[P - 1]) : (1ull << lg[0]));
        uint64_t T;
        int lp = 0, lgrp = 0;
        if (r4) {
            const uint32_t odd = lg[p] & 1u;
            lp = (int)(lg[p] + odd);
            const uint64_t cap = std::max<uint64_t>(1ull << cfg.tile_log, S << odd);
            uint64_t t_want = std::min<uint64_t>(cap / S, col_avail);
            // small transforms: prefer more, smaller tiles until the grid fills the chip four times over
            while (t_want > (1ull << odd) && batch * N / (S * t_want) < 4ull * (uint64_t)c->n_cu) t_want >>= 1;
            if (t_want < (1ull << odd)) return fail(ACX_ERR_UNSUPPORTED, "NTT plan: odd digit needs two columns");
            lgrp = r4_pick_lg(lp, (int)ilog2(t_want) - (int)odd);
            if (lgrp < 0) return fail(ACX_ERR_UNSUPPORTED, "NTT plan: no kernel instance");
            T = 1ull << (lgrp + odd);
        } else {
            T = std::min<uint64_t>(kTileElems / S, col_avail);
        }
        uint32_t no = 0;
        auto add_outer = [&](uint64_t count, uint64_t sin, uint64_t sout, uint64_t kw, uint64_t iw) {
            if (count <= 1) return;
            if (no < (uint32_t)kMaxOuter) Q.outer[no] = NttOuter{(u32)count, 0, sin, sout, kw, iw};
            ++no;
        };
        if (P == 1) {
            // columns = independent transforms of the batch
            Q.stride_t_in = Q.stride_t_out = 1;
            Q.stride_c_in = Q.stride_c_out = N;
            add_outer(batch / T, T * N, T * N, 0, 0);
        } else if (!last) {
            const uint64_t NP = 1ull << lg[P - 1];
            Q.stride_t_in = Q.stride_t_out = Wt[p];
            Q.stride_c_in = Q.stride_c_out = 1;
            Q.t_kw = Vt[p];
            const bool next_is_last = p + 1 == P - 1;
            Q.c_iw = next_is_last ? 1 : 0;
            add_outer(NP / T, T, T, 0, next_is_last ? T : 0);
            for (int q = 0; q < P - 1; ++q) {
                if (q == p) continue;
                add_outer(1ull << lg[q], Wt[q], Wt[q], q < p ? Vt[q] : 0, q == p + 1 ? Wt[q] : 0);
            }
            add_outer(batch, N, N, 0, 0);
            // twiddle w_N^(I*K), I = i_{p+1} W_{p+1}, K = k_1 + ... + k_p V_p
            uint32_t log_m = 0;
            for (int q = 0; q <= p + 1; ++q) log_m += lg[q];
            const uint32_t fold = (fold_scale && next_is_last) ? log_n : 0;
            if (log_m <= (r4 ? std::max<uint32_t>(cfg.direct_tw, 16) : 16)) {
                uint4* tw = nullptr;
                ACX_TRY(get_scaled_table(c, log_m, 1ull << log_m, inverse, fold, &tw));
                Q.tw_mode = 1; Q.tw_lo = tw; Q.tw_shift = ilog2(Wt[p + 1]);
            } else {
                uint4 *lo = nullptr, *hi = nullptr;
                ACX_TRY(get_scaled_table(c, log_n, 1024, inverse, fold, &lo));
                ACX_TRY(get_pow_table(c, log_n - 10, inverse, &hi));
                Q.tw_mode = 2; Q.tw_lo = lo; Q.tw_hi = hi; Q.tw_mask = N - 1;
            }
        } else {
            const uint64_t N1 = 1ull << lg[0];
            Q.stride_t_in = 1;            Q.stride_t_out = Vt[p];
            Q.stride_c_in = Wt[0];        Q.stride_c_out = 1;
            add_outer(N1 / T, T * Wt[0], T, 0, 0);
            for (int q = 1; q < P - 1; ++q) add_outer(1ull << lg[q], Wt[q], Vt[q], 0, 0);
            add_outer(batch, N, N, 0, 0);
        }
        if (no > (uint32_t)kMaxOuter) return fail(ACX_ERR_UNSUPPORTED, "NTT plan has too many dimensions");
        Q.n_outer = no;
        Q.log_t = ilog2(T);
        if (first && !inverse && shift_mont) Q.scale_on_load = 1;
        if (first && in_b) {
            if (Q.scale_on_load) return fail(ACX_ERR_UNSUPPORTED, "no fused product on a forward coset transform");
            Q.scale_on_load = 3;
        }
        if (last) {
            // the closing multiplication: 1 (forward), 1/N (inverse), 1/N * g^-k (inverse coset)
            // (the coset tables of an inverse transform already carry 1/N)
            const H256 s = (inverse && !shift_mont) ? hf.inv(hf.from_u64(N)) : hf.one();
            Q.scale = dev_arg(hf, s);
            Q.scale_mode = (inverse && shift_mont) ? 2 : 1;
            if (r4 && Q.scale_mode == 1 && (!inverse || fold_scale)) Q.scale_mode = 0;   // nothing left to multiply by
            if (direct_coset) Q.scale_mode = 3;                                           // one product from the direct table
            if (direct_coset && post_mont && fold_scale && post_batches > 0 && post_batches < batch) {
                Q.scale_off_end = post_batches * N;
                if (post_limited) *post_limited = true;
            }
        }
        Q.stride_t_in_hi = Q.stride_t_in;      // single stride in the transform direction (split = 0)
        Q.stride_t_out_hi = Q.stride_t_out;
        const uint64_t tiles = batch * N / (S * T);
        if (tiles > 0x7fffffffull) return fail(ACX_ERR_TOO_LARGE, "NTT grid too large");
        if (r4) {
            const bool ok = launch_ntt_r4(c->field == ACX_FIELD_BLS12_381_FR, lp, lgrp, (unsigned)tiles, cur_stream(c), Q);
            if (!ok) return fail(ACX_ERR_UNSUPPORTED, "NTT plan: kernel instance missing");
        } else {
            DISPATCH_FIELD(c, hipLaunchKernelGGL((k_ntt_tile<F>), dim3((unsigned)tiles), dim3(kBlock), 0, cur_stream(c), Q));
        }
    }
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

// A closing-factor table of one distributed step in STORE order (k_dist_table), cached per context.
//   kind 0: forward step 0 (twiddle, times g^i2 when coset != null)      kind 1: inverse step 0 (twiddle with 1/N)
//   kind 2: inverse step 1 of a coset transform (g^-(i1 C + i2); coset = 1/g)
// 32 bytes per local element (64 MB per direction for a 2^24-point transform over 8 ranks): with 288 GB of HBM that buys one
// product per element instead of two (two-level powers) and a coalesced table read instead of two dependent gathers.
int get_dist_table(acx_ctx* c, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int kind, const H256* coset, uint4** out) {
    CtxLock lock(c->mu);
    acx_ctx::DistKey key{log_n, log_r, world, rank, kind, coset ? *coset : H256{{0, 0, 0, 0}}};
    auto it = c->tw_dist.find(key);
    if (it != c->tw_dist.end()) { c->tw_dist_stamp[key] = ++c->coset_clock; *out = it->second; return ACX_OK; }
    while (c->tw_dist.size() >= acx_ctx::kDistCap) {                 // least recently used entry out
        auto victim = c->tw_dist_stamp.begin();
        for (auto s = c->tw_dist_stamp.begin(); s != c->tw_dist_stamp.end(); ++s)
            if (s->second < victim->second) victim = s;
        HIP_TRY(hipDeviceSynchronize());                             // launches that still read the table
        (void)hipFree(c->tw_dist[victim->first]);
        c->tw_dist.erase(victim->first);
        c->tw_dist_stamp.erase(victim);
    }
    const uint64_t L = (1ull << log_n) / world;
    DistTable T{};
    T.log_n = log_n; T.log_r = log_r; T.rank = rank; T.inverse = kind == 1 ? 1u : 0u; T.cols_layout = kind == 2 ? 1u : 0u;
    T.log_w = ilog2(world);
    if (kind != 2) {
        uint4 *lo = nullptr, *hi = nullptr;
        ACX_TRY(get_scaled_table(c, log_n, 1024, kind == 1, kind == 1 ? log_n : 0, &lo));
        if (log_n > 10) ACX_TRY(get_pow_table(c, log_n - 10, kind == 1, &hi));
        T.w_lo = lo; T.w_hi = hi;
    }
    if (coset) {
        uint4 *lo = nullptr, *hi = nullptr;
        ACX_TRY(get_coset_tables(c, *coset, log_n, 0, &lo, &hi));
        T.g_lo = lo; T.g_hi = hi;
    }
    uint4* tw = nullptr;
    HIP_TRY(hipMalloc((void**)&tw, L * 32));
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_dist_table<F>), dim3(grid_for(c, L)), dim3(kBlock), 0, cur_stream(c), T, tw, L));
    const hipError_t e1 = hipGetLastError(), e2 = hipStreamSynchronize(cur_stream(c));
    if (e1 != hipSuccess || e2 != hipSuccess) { (void)hipFree(tw); HIP_TRY(e1); HIP_TRY(e2); }
    c->tw_dist[key] = tw;
    c->tw_dist_stamp[key] = ++c->coset_clock;
    *out = tw;
    return ACX_OK;
}

// ---- local steps of the distributed four-step transform ------------------------------
// N = R*C, index split i = i1*C + i2, k = k1 + k2*R; W ranks; rank g owns the i2 block g (i side) and the k1
// block g (k side).  Local layouts (N/W dev elements each):
//   COLS  [i2l][i1]        x[i1*C + g*C/W + i2l]                        (i side: every local column contiguous)
//   ROWS  [kl][k2]         X[(g*R/W + kl) + k2*R]                       (k side: every local row contiguous)
//   XCHG  [peer][kl][i2l]  W contiguous chunks of (R/W)*(C/W) elements: what ONE all-to-all moves
// forward:  step 0  COLS -> XCHG  (length-R transforms over i1, times w_N^(i2*k1), coset factor s^i on load)
//           step 1  XCHG -> ROWS  (length-C transforms over i2)
// inverse:  step 0  ROWS -> XCHG  (length-C inverse transforms over k2, times w_N^-(i2*k1) / N)
//           step 1  XCHG -> COLS  (length-R inverse transforms over k1, coset factor s^-i at the end)
// Each step is ONE launch of k_ntt_r4: the transposes are strides of the pass descriptor, the twiddle is the
// kernel's closing multiplication from tables -- no separate twiddle kernel, no permute copies.
// rows_transposed (inverse step 0 only): the input is the ROWS block stored TRANSPOSED, [k2][kl] -- the ascending row order in
// which the residual kernel of a block-cyclic shard writes its dot products (mgpu.inc.h) -- instead of [kl][k2].
int ntt_dist_step_locked(acx_ctx* c, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int inverse, int step,
                         const H256* shift_mont, const uint4* in, uint4* out, bool rows_transposed = false,
                         const uint4* mul_in = nullptr, const uint4* add_out = nullptr) {
    // mul_in: the step transforms in[i] * mul_in[i] (same layout as in); add_out: out[k] = (closing step)(X[k]) + add_out[k]
    // (same layout as out) -- the h(x) pipeline's product of L and R and its coefficient-domain -O/z (qap_h_dev_locked)
    const HostField& hf = c->hf;
    const NttCfg& cfg = c->ntt;
    if ((int)log_n > hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "log_n exceeds the field's two-adicity");
    if (log_r >= log_n) return fail(ACX_ERR_INVALID_ARG, "log_r must be below log_n");
    const uint32_t log_c = log_n - log_r;
    if (log_r < 5 || log_r > 12 || log_c < 5 || log_c > 12)
        return fail(ACX_ERR_UNSUPPORTED, "distributed NTT steps need 5 <= log_r, log_n - log_r <= 12");
    if (world == 0 || (world & (world - 1)) || rank >= world) return fail(ACX_ERR_INVALID_ARG, "world must be a power of two, rank < world");
    const uint64_t N = 1ull << log_n, R = 1ull << log_r, C = 1ull << log_c;
    if (R % world || C % world) return fail(ACX_ERR_INVALID_ARG, "world must divide both factors of N");
    const uint64_t rw = R / world, cw = C / world;
    if (in == out) return fail(ACX_ERR_INVALID_ARG, "distributed NTT steps are out of place");
    // which digit this launch transforms, and over how many local columns
    const bool over_r = (step == 0) != (inverse != 0);      // forward step 0 and inverse step 1 transform the R digit
    const uint32_t ls = over_r ? log_r : log_c;
    const uint64_t S = 1ull << ls, cols = over_r ? cw : rw;
    const uint32_t odd = ls & 1u;
    const int lp = (int)(ls + odd);
    uint64_t t_want = std::min<uint64_t>(std::max<uint64_t>(1ull << cfg.tile_log, S << odd) / S, cols);
    while (t_want > (1ull << odd) && cols / t_want < 4ull * (uint64_t)c->n_cu) t_want >>= 1;
    if (t_want < (1ull << odd)) return fail(ACX_ERR_UNSUPPORTED, "distributed NTT step: odd digit needs two local columns");
    const int lgrp = r4_pick_lg(lp, (int)ilog2(t_want) - (int)odd);
    if (lgrp < 0) return fail(ACX_ERR_UNSUPPORTED, "distributed NTT step: no kernel instance");
    const uint64_t T = 1ull << (lgrp + odd);
    NttPass Q;
    std::memset(&Q, 0, sizeof(Q));
    Q.src = in; Q.dst = out;
    Q.log_s = ls; Q.log_t = ilog2(T);
    { uint4* st = nullptr; ACX_TRY(get_limb_table(c, ls, inverse, &st)); Q.sub_tw = st; }
    Q.idx_mask = N - 1;
    const uint64_t chunk = rw * cw;
    const bool twiddle_here = step == 0;
    bool xcd_outer = false;
    if (rows_transposed && !(inverse && step == 0)) return fail(ACX_ERR_INVALID_ARG, "internal: transposed input is an inverse step 0 form");
    if (!inverse && step == 0) {            // COLS -> XCHG
        Q.stride_t_in = 1; Q.stride_t_in_hi = 1; Q.stride_c_in = R;
        Q.stride_t_out = cw; Q.stride_t_out_hi = cw; Q.stride_c_out = 1;
        Q.outer[0] = NttOuter{(u32)(cols / T), 0, T * R, T, 0, T};
        Q.t_kw = 1; Q.c_iw = 1; Q.i_base = (uint64_t)rank * cw;          // K = k1, I = i2
    } else if (!inverse) {                  // XCHG -> ROWS
        Q.split_in = ilog2(cw); Q.stride_t_in = 1; Q.stride_t_in_hi = chunk; Q.stride_c_in = cw;
        Q.stride_t_out = 1; Q.stride_t_out_hi = 1; Q.stride_c_out = C;
        Q.outer[0] = NttOuter{(u32)(cols / T), 0, T * cw, T * C, 0, 0};
    } else if (step == 0 && rows_transposed) {     // ROWS^T [k2][kl] -> XCHG
        // The transform direction has stride rw and a column is 32 bytes wide, so four neighbouring columns share every
        // 128-byte line.  Tiles are therefore numbered XCD-first (workgroup b runs on XCD b % 8): an XCD's consecutive
        // workgroups take NEIGHBOURING columns, and a line is fetched from HBM into one L2 once, not into four.
        Q.stride_t_in = rw; Q.stride_t_in_hi = rw; Q.stride_c_in = 1;
        Q.split_out = ilog2(cw); Q.stride_t_out = 1; Q.stride_t_out_hi = chunk; Q.stride_c_out = cw;
        const uint64_t tiles_n = cols / T;
        if (tiles_n % 8 == 0) {
            Q.outer[0] = NttOuter{8u, 0, (tiles_n / 8) * T, (tiles_n / 8) * T * cw, 0, (tiles_n / 8) * T};
            Q.outer[1] = NttOuter{(u32)(tiles_n / 8), 0, T, T * cw, 0, T};
            xcd_outer = true;
        } else {
            Q.outer[0] = NttOuter{(u32)tiles_n, 0, T, T * cw, 0, T};
        }
        Q.t_kw = 1; Q.c_iw = 1; Q.i_base = (uint64_t)rank * rw;
    } else if (step == 0) {                 // ROWS -> XCHG
        Q.stride_t_in = 1; Q.stride_t_in_hi = 1; Q.stride_c_in = C;
        Q.split_out = ilog2(cw); Q.stride_t_out = 1; Q.stride_t_out_hi = chunk; Q.stride_c_out = cw;
        Q.outer[0] = NttOuter{(u32)(cols / T), 0, T * C, T * cw, 0, T};
        Q.t_kw = 1; Q.c_iw = 1; Q.i_base = (uint64_t)rank * rw;          // "K" = i2 (the digit), "I" = k1 (the column)
    } else {                                // XCHG -> COLS
        Q.stride_t_in = cw; Q.stride_t_in_hi = cw; Q.stride_c_in = 1;
        Q.stride_t_out = 1; Q.stride_t_out_hi = 1; Q.stride_c_out = R;
        Q.outer[0] = NttOuter{(u32)(cols / T), 0, T, T * R, 0, T};
        Q.c_iw = 1; Q.i_base = (uint64_t)rank * cw;                       // I = i2 (coset exponent only)
    }
    Q.n_outer = xcd_outer ? 2 : (Q.outer[0].count > 1 ? 1 : 0);
    Q.scale = dev_arg(hf, hf.one());
    // Closing factors come from rank-local tables in store order (get_dist_table): the twiddle w_N^(+-i2 k1) of step 0 (1/N of
    // an inverse transform folded in), and the coset factor.  The factor s^i of a forward coset transform, i = i1 C + i2,
    // splits: (s^C)^i1 depends on the transform digit only and is taken on load from a table of R entries; s^i2 is constant
    // along a column, commutes with the column's transform and rides on the store-side table.  The factor s^-i of an inverse
    // coset transform is the closing multiplication of its last step.
    const bool coset = shift_mont != nullptr;
    if (twiddle_here) {
        uint4* tw = nullptr;
        const H256* g = (!inverse && coset) ? shift_mont : nullptr;
        ACX_TRY(get_dist_table(c, log_n, log_r, world, rank, inverse ? 1 : 0, g, &tw));
        Q.tw_mode = 3; Q.tw_lo = tw;
        if (g) {
            // (s^C)^d for d < R: a direct table of the coset cache (base s^C, R entries)
            const H256 sC = hf.pow_u64(*shift_mont, C);
            uint4 *lo = nullptr, *hi = nullptr;
            ACX_TRY(get_coset_tables(c, sC, log_r, 0, &lo, &hi, 1));          // direct: all 2^log_r powers
            Q.sc_lo = lo; Q.sc_hi = nullptr;
            Q.scale_on_load = 2;
        }
    } else if (inverse && coset) {
        const H256 ginv = hf.inv(*shift_mont);
        uint4* tw = nullptr;
        ACX_TRY(get_dist_table(c, log_n, log_r, world, rank, 2, &ginv, &tw));
        Q.tw_mode = 3; Q.tw_lo = tw;
    }
    if (mul_in) {
        if (Q.scale_on_load) return fail(ACX_ERR_UNSUPPORTED, "no fused product on a forward coset step");
        Q.mul_src = mul_in; Q.scale_on_load = 3;
    }
    Q.add_src = add_out;
    const uint64_t tiles = cols / T;
    const bool ok = launch_ntt_r4(c->field == ACX_FIELD_BLS12_381_FR, lp, lgrp, (unsigned)tiles, cur_stream(c), Q);
    if (!ok) return fail(ACX_ERR_UNSUPPORTED, "distributed NTT step: kernel instance missing");
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

SellSystem sell_system(const acx_r1cs* r, const uint4* d_w, const ResidualOut& out) {
    SellSystem S;
    S.A = SellDev{r->sell_ofs[0], r->sell_tail[0], r->sell_val[0]};
    S.B = SellDev{r->sell_ofs[1], r->sell_tail[1], r->sell_val[1]};
    S.C = SellDev{r->sell_ofs[2], r->sell_tail[2], r->sell_val[2]};
    S.perm = r->perm;
    S.w = d_w;
    S.n_slices = r->n_slices;
    S.unit_c = r->unit_c ? 1u : 0u;
    S.small = r->small;
    S.out = out;
    return S;
}

// which k_r1cs_sell instance a system can run on: 0 full-width only, 1 compiled-program shape (small A and B, unit C),
// 2 mixed (per-matrix run-time flags)
inline int sell_spec(const acx_r1cs* r) {
    if (r->small == 0) return 0;
    return ((r->small & 3u) == 3u && r->unit_c) ? 1 : 2;
}
inline int sell_spec_join(int a, int b) { return a == b ? a : 2; }

// grid.x is sized by sell_grid_x for the launch's largest system: one workgroup (two waves) per slice.
inline void launch_sell(acx_ctx* c, int spec, dim3 grid, const SellSystem* systems, const SellSystem& one) {
    DISPATCH_FIELD(c, {
        if (spec == 0) hipLaunchKernelGGL((k_r1cs_sell_split<F, 0>), grid, dim3(2 * kSlice), 0, cur_stream(c), systems, one);
        else if (spec == 1) hipLaunchKernelGGL((k_r1cs_sell_split<F, 1>), grid, dim3(2 * kSlice), 0, cur_stream(c), systems, one);
        else hipLaunchKernelGGL((k_r1cs_sell_split<F, 2>), grid, dim3(2 * kSlice), 0, cur_stream(c), systems, one);
    });
}

inline unsigned sell_grid_x(uint32_t n_slices) { return ((n_slices + 7) / 8) * 8; }   // multiple of 8: the XCD remap is a bijection

// rows too long for SELL go through the CSR kernel
int launch_long_rows(acx_r1cs* r, const uint4* d_w, const ResidualOut& out, const SellSystem* d_many = nullptr, uint32_t n_many = 1) {
    acx_ctx* c = r->ctx;
    if (r->n_long == 0) return ACX_OK;
    CsrDev A{r->M[0].ptr, r->M[0].idx, r->M[0].val}, B{r->M[1].ptr, r->M[1].idx, r->M[1].val},
        C{r->M[2].ptr, r->M[2].idx, r->M[2].val};
    // long_rows holds the tiers one after the other (build_sell): <= 12, <= 24, <= 48 entries, longer
    static const uint32_t lanes[kRowTiers] = {2, 4, 8, 8};
    uint32_t first = 0;
    for (int t = 0; t < kRowTiers; ++t) {
        const uint32_t count = r->tier_rows[t];
        if (count == 0) continue;
        // many long rows: throughput matters, and eight lanes with several reductions each cost fewer instructions per row
        // than a wave with one; a few (the Split gates of a circuit) are a latency problem and take a wave per row
        const uint32_t G = (t == kRowTiers - 1 && count < 4096) ? (uint32_t)kSlice : lanes[t];
        const dim3 grid((unsigned)(((uint64_t)count * G + kBlock - 1) / kBlock), n_many, 1);
        const u32* rows = (const u32*)r->long_rows + first;
        DISPATCH_FIELD(c, {
            if (r->unit_c) hipLaunchKernelGGL((k_r1cs_residual_rows<F, true>), grid, dim3(kBlock), 0, cur_stream(c), A, B, C, d_w, rows, count, G, out, d_many);
            else hipLaunchKernelGGL((k_r1cs_residual_rows<F, false>), grid, dim3(kBlock), 0, cur_stream(c), A, B, C, d_w, rows, count, G, out, d_many);
        });
        HIP_TRY(hipGetLastError());
        first += count;
    }
    return ACX_OK;
}

int launch_residual(acx_r1cs* r, const uint4* d_w, uint64_t row_offset, unsigned long long* d_result,
                    uint4* d_res, uint4* d_dots, uint64_t dots_stride, uint32_t map_log_run = 0, uint32_t map_log_r = 0,
                    const uint4* dot_scale = nullptr) {
    acx_ctx* c = r->ctx;
    if (r->n == 0) return ACX_OK;
    const ResidualOut out{d_result, d_res, d_dots, dots_stride, row_offset, map_log_run, map_log_r, dot_scale};
    const SellSystem S = sell_system(r, d_w, out);
    const dim3 grid(sell_grid_x(r->n_slices), 1, 1);
    launch_sell(c, sell_spec(r), grid, nullptr, S);
    HIP_TRY(hipGetLastError());
    return launch_long_rows(r, d_w, out);
}

// Host side of the SELL-64 layout: row order (sorted by length inside windows), slot offsets, and
// the list of rows that stay in CSR.  Only row lengths are needed; the entries are gathered on
// the device by k_build_sell from the already converted CSR.
int build_sell(acx_r1cs* r, const uint32_t* const rowptr[3]) {
    acx_ctx* c = r->ctx;
    const uint64_t n = r->n;
    const uint32_t n_slices = (uint32_t)((n + kSlice - 1) / kSlice);
    r->n_slices = n_slices;
    if (n == 0) return ACX_OK;
    PhaseTimer pt;
    std::vector<uint32_t> key(n), perm((size_t)n_slices * kSlice, kNoRow), longs, tiers[kRowTiers];
    std::vector<uint32_t> ofs[3];
    StreamDrain drain(cur_stream(c));          // after the vectors above: they outlive every copy enqueued from them
    // Row classes: (lenA, lenB, lenC) with every length <= kSellMaxLen, or "long".  Few classes, so the stable sort of a
    // window is a counting sort (a comparison sort of 2^20 rows cost 32 ms of a 110 ms load).
    constexpr uint32_t kLenRadix = kSellMaxLen + 1, kLongClass = kLenRadix * kLenRadix * kLenRadix;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t l[3];
        bool is_long = false;
        for (int k = 0; k < 3; ++k) { l[k] = rowptr[k][i + 1] - rowptr[k][i]; is_long = is_long || l[k] > (uint32_t)kSellMaxLen; }
        key[i] = is_long ? kLongClass : (l[0] * kLenRadix + l[1]) * kLenRadix + l[2];
        if (is_long) {
            const uint32_t mx = std::max({l[0], l[1], l[2]});
            tiers[mx <= 2 * kWideTerms ? 0 : mx <= 4 * kWideTerms ? 1 : mx <= 8 * kWideTerms ? 2 : 3].push_back((uint32_t)i);
        }
    }
    for (int t = 0; t < kRowTiers; ++t) {
        r->tier_rows[t] = (uint32_t)tiers[t].size();
        longs.insert(longs.end(), tiers[t].begin(), tiers[t].end());
    }
    std::vector<uint32_t> start(kLongClass + 2);
    for (uint64_t ws = 0; ws < n; ws += kSellWindow) {
        const uint64_t we = std::min<uint64_t>(ws + kSellWindow, n);
        std::fill(start.begin(), start.end(), 0u);
        for (uint64_t i = ws; i < we; ++i) ++start[key[i] + 1];
        for (uint32_t k = 0; k <= kLongClass; ++k) start[k + 1] += start[k];
        for (uint64_t i = ws; i < we; ++i)                       // ascending class, original order inside a class
            perm[ws + start[key[i]]++] = key[i] == kLongClass ? kNoRow : (uint32_t)i;
    }
    pt.mark("  sell: keys + window sorts");
    r->n_long = (uint32_t)longs.size();
    uint64_t slots[3];
    for (int k = 0; k < 3; ++k) {
        ofs[k].resize(n_slices + 1);
        ofs[k][0] = 0;
        for (uint32_t s = 0; s < n_slices; ++s) {
            uint32_t mx = 0;
            for (int l = 0; l < kSlice; ++l) {
                const uint32_t row = perm[(size_t)s * kSlice + l];
                if (row != kNoRow) mx = std::max(mx, rowptr[k][row + 1] - rowptr[k][row]);
            }
            ofs[k][s + 1] = ofs[k][s] + mx;
        }
        slots[k] = ofs[k][n_slices];
    }
    pt.mark("  sell: slice offsets");
    {   // one allocation for everything the SELL form holds
        size_t off = 0, o_ofs[3], o_tail[3], o_val[3];
        const size_t o_perm = off; off += align256(perm.size() * 4);
        const size_t o_long = off; off += align256(std::max<size_t>(longs.size(), 1) * 4);
        for (int k = 0; k < 3; ++k) {
            o_ofs[k] = off; off += align256(ofs[k].size() * 4);
            o_tail[k] = off; off += align256(std::max<uint64_t>(slots[k], 1) * kSlice * 8);
            o_val[k] = off;
            if (!((r->small >> k) & 1u)) off += align256(std::max<uint64_t>(slots[k], 1) * kSlice * 32);
        }
        if (hipMalloc(&r->sell_slab, off) != hipSuccess) { (void)hipGetLastError(); r->sell_slab = nullptr; return fail(ACX_ERR_OOM, "device allocation failed"); }
        uint8_t* base = static_cast<uint8_t*>(r->sell_slab);
        r->perm = (u32*)(base + o_perm);
        if (!longs.empty()) r->long_rows = (u32*)(base + o_long);
        for (int k = 0; k < 3; ++k) {
            r->sell_ofs[k] = (u32*)(base + o_ofs[k]);
            r->sell_tail[k] = (uint2*)(base + o_tail[k]);
            if (!((r->small >> k) & 1u)) r->sell_val[k] = (uint4*)(base + o_val[k]);
        }
    }
    HIP_TRY(hipMemcpyAsync(r->perm, perm.data(), perm.size() * 4, hipMemcpyHostToDevice, cur_stream(c)));
    if (!longs.empty()) HIP_TRY(hipMemcpyAsync(r->long_rows, longs.data(), longs.size() * 4, hipMemcpyHostToDevice, cur_stream(c)));
    // the device's check of the small-coefficient classification: one flag for the three matrices, fetched after the last launch
    uint32_t* d_bad = nullptr;
    if (r->small) {
        d_bad = cur_err(c) + 1;                      // second pad word of the call's result slot (the first is the canonicity flag)
        HIP_TRY(hipMemsetAsync(d_bad, 0, 4, cur_stream(c)));
    }
    for (int k = 0; k < 3; ++k) {
        HIP_TRY(hipMemcpyAsync(r->sell_ofs[k], ofs[k].data(), ofs[k].size() * 4, hipMemcpyHostToDevice, cur_stream(c)));
        const CsrDev M{r->M[k].ptr, r->M[k].idx, r->M[k].val};
        if ((r->small >> k) & 1u) {
            DISPATCH_FIELD(c, hipLaunchKernelGGL((k_build_sell_small<F>), dim3((n_slices + 3) / 4), dim3(kBlock), 0, cur_stream(c), M,
                                                 (const u32*)r->perm, (const u32*)r->sell_ofs[k], n_slices, r->sell_tail[k], d_bad));
        } else {
            hipLaunchKernelGGL(k_build_sell, dim3((n_slices + 3) / 4), dim3(kBlock), 0, cur_stream(c), M, (const u32*)r->perm,
                               (const u32*)r->sell_ofs[k], n_slices, r->sell_tail[k], r->sell_val[k]);
        }
        HIP_TRY(hipGetLastError());
    }
    uint32_t bad = 0;
    if (d_bad) HIP_TRY(hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, cur_stream(c)));
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));           // perm / longs / ofs (and the caller's matrices) are read by copies until here
    pt.mark("  sell: device build");
    if (bad) return fail(ACX_ERR_HIP, "small-coefficient classification disagrees with the device");
    return ACX_OK;
}

// canonical value v with v <= 2^27 or p - v <= 2^27 (kSmallCoeffMax)
bool is_small_coeff(const HostField& hf, const acx_fr& f) {
    H256 v;
    std::memcpy(v.l, f.b, 32);
    if ((v.l[1] | v.l[2] | v.l[3]) == 0 && v.l[0] <= (uint64_t)kSmallCoeffMax) return true;
    const H256& p = hf.modulus();
    uint64_t d[4];
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        const unsigned __int128 t = (unsigned __int128)p.l[i] - v.l[i] - (uint64_t)borrow;
        d[i] = (uint64_t)t;
        borrow = (t >> 64) & 1;
    }
    return borrow == 0 && (d[1] | d[2] | d[3]) == 0 && d[0] <= (uint64_t)kSmallCoeffMax;
}


// Sort + merge duplicate columns of one host CSR row set (only rows that need it).
int normalise_csr(const HostField& hf, uint64_t n, uint64_t m, const acx_csr* in, std::vector<uint32_t>& rowptr,
                  std::vector<uint32_t>& col, std::vector<acx_fr>& val) {
    if (!in || !in->rowptr) return fail(ACX_ERR_INVALID_ARG, "null CSR");
    const uint64_t nnz = in->rowptr[n];
    if (in->rowptr[0] != 0) return fail(ACX_ERR_INVALID_ARG, "rowptr[0] != 0");
    if (nnz && (!in->col || !in->val)) return fail(ACX_ERR_INVALID_ARG, "null CSR arrays");
    rowptr.assign(1, 0);
    rowptr.reserve(n + 1);
    col.reserve(nnz);
    val.reserve(nnz);
    std::vector<std::pair<uint32_t, uint64_t>> tmp;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t e0 = in->rowptr[i], e1 = in->rowptr[i + 1];
        if (e1 < e0 || e1 > nnz) return fail(ACX_ERR_INVALID_ARG, "rowptr not monotone");
        bool sorted = true;
        for (uint32_t e = e0; e < e1; ++e) {
            if (in->col[e] >= m) return fail(ACX_ERR_INVALID_ARG, "column index >= m");
            if (e > e0 && in->col[e] <= in->col[e - 1]) sorted = false;
        }
        if (sorted) {
            col.insert(col.end(), in->col + e0, in->col + e1);
            val.insert(val.end(), in->val + e0, in->val + e1);
        } else {
            tmp.clear();
            for (uint32_t e = e0; e < e1; ++e) tmp.emplace_back(in->col[e], e);
            std::stable_sort(tmp.begin(), tmp.end(), [](auto& a, auto& b) { return a.first < b.first; });
            for (size_t k = 0; k < tmp.size();) {
                H256 acc;
                ACX_TRY(read_h256(&in->val[tmp[k].second], hf, acc));
                size_t j = k + 1;
                for (; j < tmp.size() && tmp[j].first == tmp[k].first; ++j) {
                    H256 t;
                    ACX_TRY(read_h256(&in->val[tmp[j].second], hf, t));
                    acc = hf.add(acc, t);
                }
                col.push_back(tmp[k].first);
                acx_fr f;
                write_h256(&f, hf, acc);
                val.push_back(f);
                k = j;
            }
        }
        rowptr.push_back((uint32_t)col.size());
    }
    return ACX_OK;
}

// Enqueue the upload of one CSR matrix into buffers the caller carved out of the system's slab (out.ptr / idx / val set), values
// converted to dev format in place.  No wait: a non-canonical value raises the flag of the call's result slot (begin_call /
// end_call_fetch), and the host arrays must stay alive until the caller has synchronised the stream.
int upload_matrix_async(acx_ctx* c, const uint32_t* ptr, size_t n_ptr, const uint32_t* idx, size_t nnz, const acx_fr* val, DevMatrix& out) {
    out.nnz = nnz;
    HIP_TRY(hipMemcpyAsync(out.ptr, ptr, n_ptr * 4, hipMemcpyHostToDevice, cur_stream(c)));
    if (nnz) HIP_TRY(hipMemcpyAsync(out.idx, idx, nnz * 4, hipMemcpyHostToDevice, cur_stream(c)));
    return upload_elements_async(c, val, nnz, out.val);
}

void free_matrix(DevMatrix& mtx) {
    if (mtx.ptr) (void)hipFree(mtx.ptr);
    if (mtx.idx) (void)hipFree(mtx.idx);
    if (mtx.val) (void)hipFree(mtx.val);
    if (mtx.colid) (void)hipFree(mtx.colid);
    mtx = DevMatrix{};
}

// the column views: the slab when build_csc made them, member by member otherwise
void free_csc(acx_r1cs* r) {
    if (r->csc_slab) {
        (void)hipFree(r->csc_slab);
        r->csc_slab = nullptr;
        for (int k = 0; k < 3; ++k) { r->T[k].ptr = nullptr; r->T[k].idx = nullptr; r->T[k].colid = nullptr; r->T[k].val = nullptr; }
    }
    for (int k = 0; k < 3; ++k) free_matrix(r->T[k]);
}

void free_r1cs_device(acx_r1cs* r) {
    free_csc(r);
    if (r->slab) {                                   // the members below are views of the two slabs
        (void)hipFree(r->slab);
        r->slab = nullptr;
        for (int k = 0; k < 3; ++k) { r->M[k].ptr = nullptr; r->M[k].idx = nullptr; r->M[k].val = nullptr; }
        r->d_w = nullptr; r->d_hscale = nullptr;
    }
    if (r->sell_slab) {
        (void)hipFree(r->sell_slab);
        r->sell_slab = nullptr;
        for (int k = 0; k < 3; ++k) { r->sell_ofs[k] = nullptr; r->sell_tail[k] = nullptr; r->sell_val[k] = nullptr; }
        r->perm = nullptr; r->long_rows = nullptr;
    }
    for (int k = 0; k < 3; ++k) {
        free_matrix(r->M[k]);
        if (r->sell_ofs[k]) (void)hipFree(r->sell_ofs[k]);
        if (r->sell_tail[k]) (void)hipFree(r->sell_tail[k]);
        if (r->sell_val[k]) (void)hipFree(r->sell_val[k]);
        r->sell_ofs[k] = nullptr; r->sell_tail[k] = nullptr; r->sell_val[k] = nullptr;
    }
    if (r->perm) (void)hipFree(r->perm);
    if (r->long_rows) (void)hipFree(r->long_rows);
    if (r->ev_mul) { (void)hipFree(r->ev_mul); r->ev_mul = nullptr; }
    if (r->ev_cols) { (void)hipFree(r->ev_cols); r->ev_cols = nullptr; }
    if (r->ev_equal) { (void)hipFree(r->ev_equal); r->ev_equal = nullptr; }
    if (r->ev_level_ofs) { (void)hipFree(r->ev_level_ofs); r->ev_level_ofs = nullptr; }
    if (r->d_w_canon) { (void)hipFree(r->d_w_canon); r->d_w_canon = nullptr; }
    if (r->ev_items) (void)hipFree(r->ev_items);
    if (r->ev_row) (void)hipFree(r->ev_row);
    if (r->ev_wire_ofs) (void)hipFree(r->ev_wire_ofs);
    if (r->ev_wires) (void)hipFree(r->ev_wires);
    if (r->ev_kind) (void)hipFree(r->ev_kind);
    r->ev_items = r->ev_row = r->ev_wire_ofs = r->ev_wires = nullptr; r->ev_kind = nullptr;
    if (r->d_w) (void)hipFree(r->d_w);
    if (r->qh) (void)hipFree(r->qh);
    if (r->d_hscale) (void)hipFree(r->d_hscale);
    r->perm = nullptr; r->long_rows = nullptr; r->d_w = nullptr; r->qh = nullptr; r->d_hscale = nullptr;
}

int r1cs_from_host(acx_ctx* ctx, uint64_t n, uint64_t m, const acx_csr* const mats[3], acx_r1cs** out) {
    if (!ctx || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (m == 0 || m >= 0xffffffffull || n >= 0xffffffffull) return fail(ACX_ERR_TOO_LARGE, "n or m out of range");
    const uint32_t log_n = ceil_log2(std::max<uint64_t>(n, 1));
    if ((int)log_n > ctx->hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "n exceeds 2^two_adicity");
    CtxLock lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    acx_r1cs* r = new (std::nothrow) acx_r1cs();
    if (!r) return fail(ACX_ERR_OOM, "host allocation failed");
    r->ctx = ctx; r->n = n; r->m = m; r->log_n = log_n;
    int rc = ACX_OK;
    PhaseTimer pt;
    try {                                              // host vectors are sized by caller data
        // A matrix whose rows arrive sorted by column without duplicates (what every producer in this repository and the
        // Haskell marshaller emit) is used in place: validated by worker threads, uploaded straight from the caller's
        // arrays.  Anything else is sorted / merged into a private copy first.
        std::vector<uint32_t> own_rowptr[3], own_col[3];
        std::vector<acx_fr> own_val[3];
        const uint32_t* rowptrs[3] = {nullptr, nullptr, nullptr};
        const uint32_t* cols[3] = {nullptr, nullptr, nullptr};
        const acx_fr* vals[3] = {nullptr, nullptr, nullptr};
        uint64_t nnzs[3] = {0, 0, 0};
        for (int k = 0; k < 3 && rc == ACX_OK; ++k) {
            const acx_csr* in = mats[k];
            if (!in || !in->rowptr) { rc = fail(ACX_ERR_INVALID_ARG, "null CSR"); break; }
            const uint32_t* rowptr = in->rowptr;
            const uint32_t* col = in->col;
            const acx_fr* val = in->val;
            const uint64_t nnz_in = in->rowptr[n];
            if (in->rowptr[0] != 0) { rc = fail(ACX_ERR_INVALID_ARG, "rowptr[0] != 0"); break; }
            if (nnz_in && (!in->col || !in->val)) { rc = fail(ACX_ERR_INVALID_ARG, "null CSR arrays"); break; }
            std::atomic<int> state{0};                         // 0 in place, 1 needs normalising, 2 invalid (reported by normalise_csr)
            parallel_ranges(n, host_threads(n, 1 << 16), [&](unsigned, uint64_t b, uint64_t e) {
                for (uint64_t i = b; i < e && state.load(std::memory_order_relaxed) == 0; ++i) {
                    const uint32_t e0 = rowptr[i], e1 = rowptr[i + 1];
                    if (e1 < e0 || e1 > nnz_in) { state = 2; return; }
                    for (uint32_t q = e0; q < e1; ++q) {
                        if (col[q] >= m) { state = 2; return; }
                        if (q > e0 && col[q] <= col[q - 1]) { state = 1; return; }
                    }
                }
            });
            if (state != 0) {
                rc = normalise_csr(ctx->hf, n, m, in, own_rowptr[k], own_col[k], own_val[k]);
                rowptr = own_rowptr[k].data(); col = own_col[k].data(); val = own_val[k].data();
            }
            rowptrs[k] = rowptr;
            const uint64_t nnz = rc == ACX_OK ? rowptr[n] : 0;
            pt.mark("validate / normalise");
            if (rc == ACX_OK && k == 2) {
                static const uint8_t one32[32] = {1};
                bool unit = true;
                for (uint64_t e = 0; e < nnz && unit; ++e) unit = std::memcmp(val[e].b, one32, 32) == 0;
                r->unit_c = unit;
            }
            // small-coefficient form (kernels.hip.h sell_dot_small): every entry of the rows this matrix keeps in SELL
            // is c or p - c with c <= 2^27.  Rows longer than the SELL cut-over go through the CSR kernel whatever
            // they hold (Split gates: powers of two up to 2^255), so they do not count.
            if (rc == ACX_OK && ctx->small_coeff && !(k == 2 && r->unit_c)) {
                bool small = nnz != 0;
                for (uint64_t i = 0; i < n && small; ++i) {
                    const uint32_t e0 = rowptr[i], e1 = rowptr[i + 1];
                    if (e1 - e0 > (uint32_t)kSellMaxLen) continue;
                    for (uint32_t e = e0; e < e1 && small; ++e) small = is_small_coeff(ctx->hf, val[e]);
                }
                if (small) r->small |= 1u << k;
            }
            pt.mark("classify");
            cols[k] = col; vals[k] = val; nnzs[k] = nnz;
        }
        // one allocation for the three matrices, the resident witness and the h(x) constants; every upload enqueued without a
        // wait, ONE canonicity flag for all values (the call's result slot), one stream wait at the end of build_sell
        const bool with_h = (int)log_n + 1 <= ctx->hf.two_adicity();
        H256 hpair[2];
        if (rc == ACX_OK) {
            size_t off = 0, o_ptr[3], o_idx[3], o_val[3];
            for (int k = 0; k < 3; ++k) {
                o_ptr[k] = off; off += align256((n + 1) * 4);
                o_idx[k] = off; off += align256(std::max<uint64_t>(nnzs[k], 1) * 4);
                o_val[k] = off; off += align256(std::max<uint64_t>(nnzs[k], 1) * 32);
            }
            const size_t o_w = off; off += align256(m * 32);
            const size_t o_h = off; off += 256;
            if (hipMalloc(&r->slab, off) != hipSuccess) { (void)hipGetLastError(); r->slab = nullptr; rc = fail(ACX_ERR_OOM, "device allocation failed"); }
            if (rc == ACX_OK) {
                uint8_t* base = static_cast<uint8_t*>(r->slab);
                for (int k = 0; k < 3; ++k) {
                    r->M[k].ptr = (u32*)(base + o_ptr[k]); r->M[k].idx = (u32*)(base + o_idx[k]); r->M[k].val = (uint4*)(base + o_val[k]);
                }
                r->d_w = (uint4*)(base + o_w);
                rc = begin_call(ctx);
                for (int k = 0; k < 3 && rc == ACX_OK; ++k) rc = upload_matrix_async(ctx, rowptrs[k], n + 1, cols[k], nnzs[k], vals[k], r->M[k]);
                if (rc == ACX_OK && with_h) {
                    // {1/z, -1/z}, z = g^N - 1 (the target polynomial on the coset g<omega>): the factors the h(x) pipeline lets
                    // ride on the stored dot products.  They depend on N alone; made here so that concurrent callers find them ready.
                    const HostField& hf = ctx->hf;
                    const H256 zinv = hf.inv(hf.sub(hf.pow_u64(hf.generator(), 1ull << log_n), hf.one()));
                    hpair[0] = hf.to_dev_word(zinv); hpair[1] = hf.to_dev_word(hf.sub(hf.zero(), zinv));
                    r->d_hscale = (uint4*)(base + o_h);
                    if (hipMemcpyAsync(r->d_hscale, hpair, 64, hipMemcpyHostToDevice, cur_stream(ctx)) != hipSuccess) rc = fail(ACX_ERR_HIP, "h(x) constants");
                }
            }
            pt.mark("upload (enqueued)");
        }
        if (rc == ACX_OK) rc = build_sell(r, rowptrs);                     // ends with the stream wait: host arrays are free after it
        else if (r->slab) (void)hipStreamSynchronize(cur_stream(ctx));    // never leave copies from host arrays in flight
        pt.mark("build_sell");
        if (rc == ACX_OK) {
            CallSlot& slot = cur_hslot(ctx);
            rc = end_call_fetch(ctx, &slot);
            if (rc == ACX_OK && hipStreamSynchronize(cur_stream(ctx)) != hipSuccess) rc = fail(ACX_ERR_HIP, "stream");
            if (rc == ACX_OK && slot.noncanonical) rc = fail(ACX_ERR_NONCANONICAL, "element >= p");
        }
    } catch (const std::bad_alloc&) {
        rc = fail(ACX_ERR_OOM, "host allocation failed");
    }
    if (rc != ACX_OK) {
        free_r1cs_device(r);
        delete r;
        return rc;
    }
    *out = r;
    return ACX_OK;
}

// Build the CSC copies on the device from the device CSR: histogram, scan, fill (kernels.hip.h K6).
static int build_csc(acx_r1cs* r) {
    acx_ctx* c = r->ctx;
    const hipStream_t st = cur_stream(c);
    // one allocation for the three views and the histogram / cursor scratch (the launches of the three matrices are ordered on
    // one stream, so they share the scratch), one wait at the end: 18 hipMallocs, 6 hipFrees and 3 waits before
    size_t off = 0, o_ptr[3], o_idx[3], o_colid[3], o_val[3];
    for (int k = 0; k < 3; ++k) {
        const uint64_t e = std::max<uint64_t>(r->M[k].nnz, 1);
        o_ptr[k] = off; off += align256((r->m + 1) * 4);
        o_idx[k] = off; off += align256(e * 4);
        o_colid[k] = off; off += align256(e * 4);
        o_val[k] = off; off += align256(e * 32);
    }
    const size_t o_count = off; off += align256((r->m + 1) * 4);
    const size_t o_cursor = off; off += align256((r->m + 1) * 4);
    if (hipMalloc(&r->csc_slab, off) != hipSuccess) { (void)hipGetLastError(); r->csc_slab = nullptr; return fail(ACX_ERR_OOM, "device allocation failed"); }
    uint8_t* base = static_cast<uint8_t*>(r->csc_slab);
    u32* count = (u32*)(base + o_count);
    u32* cursor = (u32*)(base + o_cursor);
    for (int k = 0; k < 3; ++k) {
        const DevMatrix& M = r->M[k];
        DevMatrix& T = r->T[k];
        T.nnz = M.nnz;
        T.ptr = (u32*)(base + o_ptr[k]); T.idx = (u32*)(base + o_idx[k]); T.colid = (u32*)(base + o_colid[k]); T.val = (uint4*)(base + o_val[k]);
        HIP_TRY(hipMemsetAsync(count, 0, (r->m + 1) * 4, st));
        HIP_TRY(hipMemsetAsync(cursor, 0, (r->m + 1) * 4, st));
        if (M.nnz) hipLaunchKernelGGL(k_col_histogram, dim3(grid_for(c, M.nnz)), dim3(kBlock), 0, st, (const u32*)M.idx, M.nnz, count);
        hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, st, (const u32*)count, T.ptr, r->m);
        if (M.nnz) {
            const CsrDev csr{M.ptr, M.idx, M.val};
            hipLaunchKernelGGL(k_csc_fill, dim3(grid_for(c, r->n)), dim3(kBlock), 0, st, csr, r->n, (const u32*)T.ptr, cursor,
                               T.idx, T.colid, T.val);
        }
        HIP_TRY(hipGetLastError());
        T.h_ptr.resize(r->m + 1);                  // host copy of colptr: qap_columns_core sorts a batch into sparse and dense columns with it
        HIP_TRY(hipMemcpyAsync(T.h_ptr.data(), T.ptr, (r->m + 1) * 4, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(hipStreamSynchronize(st));             // the host copies are complete; other lanes may use the views from here on
    return ACX_OK;
}

// Build the CSC copies on the device from the device CSR: histogram, scan, fill (kernels.hip.h K6).  All three or none: a
// failure part-way (device OOM on the third matrix) releases what the earlier ones allocated, so a retry starts clean.
int ensure_csc(acx_r1cs* r) {
    acx_ctx* c = r->ctx;
    CtxLock lock(c->mu);
    if (r->has_csc) return ACX_OK;
    const int rc = build_csc(r);
    if (rc != ACX_OK) {
        (void)hipStreamSynchronize(cur_stream(c));
        free_csc(r);
        return rc;
    }
    r->has_csc = true;
    return ACX_OK;
}

// createPolynomialsFFT for wires [wire_begin, wire_begin + cnt) of one matrix, on the calling thread's stream:
// d_out (cnt * N dev elements) receives the coefficients, d_len (cnt) the stripped lengths.
int qap_columns_core(acx_r1cs* r, int matrix, uint64_t wire_begin, uint64_t cnt, uint4* d_out, unsigned long long* d_len) {
    acx_ctx* c = r->ctx;
    const uint64_t N = 1ull << r->log_n;
    const DevMatrix& T = r->T[matrix];
    if (cnt == 0) return ACX_OK;
    // Columns of at most kDirectMid entries (nearly every wire of a gate-list circuit) are interpolated directly
    // (k_col_direct up to 4 entries, k_col_direct_mid for 5 .. 8: k products per coefficient); the others -- inputs used by many gates, the constant wire -- form
    // runs that take the batched inverse transform.  Many short runs: the whole batch takes the transform.
    static const bool direct_ok = [] { const char* e = getenv("ACX_COLUMNS_DIRECT"); return !e || atoi(e) != 0; }();
    std::vector<std::pair<uint64_t, uint64_t>> runs;      // dense runs [begin, end) inside the batch
    uint64_t n_sparse = 0, n_mid = 0;
    if (direct_ok && T.h_ptr.size() > wire_begin + cnt) {
        const uint32_t* hp = T.h_ptr.data() + wire_begin;
        for (uint64_t i = 0; i < cnt; ++i) {
            if (hp[i + 1] - hp[i] <= kDirectMid) { ++n_sparse; n_mid += hp[i + 1] - hp[i] > kDirectMax; continue; }
            if (!runs.empty() && runs.back().second == i) runs.back().second = i + 1; else runs.emplace_back(i, i + 1);
        }
    }
    if (n_sparse == 0 || runs.size() > 16) { runs.assign(1, {0, cnt}); n_sparse = 0; n_mid = 0; }
    for (const auto& run : runs)
        HIP_TRY(hipMemsetAsync(d_out + 2 * run.first * N, 0, (run.second - run.first) * N * 32, cur_stream(c)));
    if (T.nnz && !runs.empty())       // entries of sparse columns land in memory the direct kernel overwrites afterwards
        hipLaunchKernelGGL(k_scatter_columns, dim3(grid_for(c, T.nnz / 4 + 1)), dim3(kBlock), 0, cur_stream(c), (const u32*)T.ptr, (const u32*)T.idx,
                           (const u32*)T.colid, (const uint4*)T.val, wire_begin, cnt, r->log_n, d_out);
    for (const auto& run : runs)
        ACX_TRY(ntt_dev_locked(c, d_out + 2 * run.first * N, r->log_n, run.second - run.first, 1, nullptr));
    if (n_sparse) {
        ColDirect P{};
        P.colptr = T.ptr; P.rowidx = T.idx; P.val = T.val; P.log_n = r->log_n;
        P.steps = (u32)std::max<uint64_t>(1, std::min<uint64_t>(32, N / kBlock));
        uint4 *lo = nullptr, *hi = nullptr, *blk = nullptr;
        ACX_TRY(get_low_table(c, r->log_n, 1, &lo));
        if (r->log_n > 10) ACX_TRY(get_pow_table(c, r->log_n - 10, 1, &hi));
        ACX_TRY(get_pow_table(c, r->log_n > 8 ? r->log_n - 8 : 0, 1, &blk));     // omega_N^-(256 j): the block / step factors, read by scalar loads
        P.tw_lo = lo; P.tw_hi = hi; P.tw_blk = blk;
        P.inv_n = dev_arg(c->hf, c->hf.inv(c->hf.from_u64(N)));
        const unsigned gx = (unsigned)std::max<uint64_t>(1, N / ((uint64_t)kBlock * P.steps));
        for (uint64_t b = 0; b < cnt; b += 32768) {
            const uint64_t nb = std::min<uint64_t>(32768, cnt - b);
            P.wire_begin = wire_begin + b;
            DISPATCH_FIELD(c, hipLaunchKernelGGL((k_col_direct<F>), dim3(gx, (unsigned)nb), dim3(kBlock), 0, cur_stream(c), P, d_out + 2 * b * N));
            if (n_mid)      // columns of 5 .. 8 entries in the batch: the same grid once more, every other block leaves at once
                DISPATCH_FIELD(c, hipLaunchKernelGGL((k_col_direct_mid<F>), dim3(gx, (unsigned)nb), dim3(kBlock), 0, cur_stream(c), P, d_out + 2 * b * N));
        }
    }
    if (d_len) DISPATCH_FIELD(c, hipLaunchKernelGGL((k_poly_len<F>), dim3((unsigned)cnt), dim3(kBlock), 0, cur_stream(c), (const uint4*)d_out, r->log_n, d_len, (const u32*)T.ptr + wire_begin));
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

}  // namespace

int qap_columns_host(acx_r1cs* r, int matrix, uint64_t wire_begin, uint64_t wire_count, acx_fr* out, uint64_t* out_len,
                     uint64_t max_batch_bytes);

// ==================================================================================== C ABI
extern "C" {

#include "circuit_abi.inc.h"      // acx_strerror .. acx_circuit_rows_lists: pure host code, shared with host_only.cpp

int acx_ctx_create(int field, int device_id, acx_ctx** out) {
    if (!out) return fail(ACX_ERR_INVALID_ARG, "null out pointer");
    if (field != ACX_FIELD_BN254_FR && field != ACX_FIELD_BLS12_381_FR) return fail(ACX_ERR_INVALID_ARG, "unknown field");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(ACX_ERR_NO_DEVICE, "no HIP device visible (libacx has no CPU fallback)");
    if (device_id < 0 || device_id >= count) return fail(ACX_ERR_NO_DEVICE, "device id out of range");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device_id));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return fail(ACX_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", libacx is built for gfx950 only");
    HIP_TRY(hipSetDevice(device_id));
    acx_ctx* c = new (std::nothrow) acx_ctx();
    if (!c) return fail(ACX_ERR_OOM, "host allocation failed");
    c->field = field;
    c->device = device_id;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->hf = field == ACX_FIELD_BN254_FR ? HostField::make<Bn254Fr>() : HostField::make<Bls12381Fr>();
    c->ntt = ntt_cfg_from_env();
    if (const char* e = std::getenv("ACX_R1CS_SMALL")) c->small_coeff = std::atoi(e) != 0;   // development A/B switch
    bool ok = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) == hipSuccess &&
              hipMalloc((void**)&c->d_result, 32) == hipSuccess &&    // {n_bad, first_bad, canonicity flag, pad}: one copy in, one out
              hipHostMalloc(&c->h_slot, 64) == hipSuccess;
    if (ok) c->d_err = (uint32_t*)(c->d_result + 2);
    for (auto& ln : c->lanes)
        ok = ok && hipStreamCreateWithFlags(&ln.stream, hipStreamNonBlocking) == hipSuccess &&
             hipMalloc((void**)&ln.d_result, 32) == hipSuccess && hipHostMalloc(&ln.h_slot, 64) == hipSuccess;
    if (ok) for (auto& ln : c->lanes) ln.d_err = (uint32_t*)(ln.d_result + 2);
    if (!ok) {
        acx_ctx_destroy(c);
        return fail(ACX_ERR_HIP, "context resource creation failed");
    }
    *out = c;
    return ACX_OK;
}

void acx_ctx_destroy(acx_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    for (auto& kv : c->twiddles) (void)hipFree(kv.second);
    for (auto& kv : c->tw_low) (void)hipFree(kv.second);
    for (auto& kv : c->tw_scaled) (void)hipFree(kv.second);
    for (auto& kv : c->tw_limbs) (void)hipFree(kv.second);
    for (auto& kv : c->tw_dist) (void)hipFree(kv.second);
    for (auto& kv : c->h_scale) (void)hipFree(kv.second);
    if (c->ntt_scratch) (void)hipFree(c->ntt_scratch);
    for (auto& e : c->cosets) { if (e.lo) (void)hipFree(e.lo); if (e.hi) (void)hipFree(e.hi); }
    c->cosets.clear();
    if (c->d_result) (void)hipFree(c->d_result);                   // d_err lives inside it
    if (c->h_slot) (void)hipHostFree(c->h_slot);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    for (auto& ln : c->lanes) {
        if (ln.d_result) (void)hipFree(ln.d_result);
        if (ln.h_slot) (void)hipHostFree(ln.h_slot);
        if (ln.arena) (void)hipFree(ln.arena);
        if (ln.ntt_scratch) (void)hipFree(ln.ntt_scratch);
        for (auto& e : ln.ev) if (e) (void)hipEventDestroy(e);
        if (ln.copy_stream) (void)hipStreamDestroy(ln.copy_stream);
        if (ln.stream) (void)hipStreamDestroy(ln.stream);
    }
    delete c;
}

int acx_ctx_set_root(acx_ctx* c, uint32_t two_adicity, const acx_fr* omega) {
    if (!c || !omega || two_adicity == 0 || two_adicity > 64) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    CtxLock lock(c->mu);
    H256 w;
    ACX_TRY(read_h256(omega, c->hf, w));
    // must have exact order 2^two_adicity: w^(2^(s-1)) == -1
    H256 t = w;
    for (uint32_t i = 0; i + 1 < two_adicity; ++i) t = c->hf.mul(t, t);
    if (t != c->hf.neg(c->hf.one())) return fail(ACX_ERR_INVALID_ARG, "omega is not a primitive 2^s-th root of unity");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipDeviceSynchronize());                       // nothing in flight may still read the old tables
    for (auto& kv : c->twiddles) (void)hipFree(kv.second);
    for (auto& kv : c->tw_low) (void)hipFree(kv.second);
    for (auto& kv : c->tw_scaled) (void)hipFree(kv.second);
    for (auto& kv : c->tw_limbs) (void)hipFree(kv.second);
    for (auto& kv : c->tw_dist) (void)hipFree(kv.second);
    c->tw_dist.clear();
    c->tw_dist_stamp.clear();
    c->twiddles.clear();
    c->tw_low.clear();
    c->tw_scaled.clear();
    c->tw_limbs.clear();
    c->hf.set_omega_max(w, (int)two_adicity);
    return ACX_OK;
}

int acx_ctx_root_of_unity(acx_ctx* c, uint32_t k, acx_fr* out) {
    if (!c || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if ((int)k > c->hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "getRootOfUnity: exponent out of range");
    write_h256(out, c->hf, c->hf.root_of_unity((int)k));
    return ACX_OK;
}

int acx_ctx_sync(acx_ctx* c) {
    if (!c) return fail(ACX_ERR_INVALID_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto& ln : c->lanes) HIP_TRY(hipStreamSynchronize(ln.stream));
    return ACX_OK;
}

void* acx_ctx_stream(acx_ctx* c) { return c ? (void*)c->stream : nullptr; }

static int circuit_to_r1cs_impl(acx_ctx* ctx, const acx_circuit* c, const acx_fr* roots, uint64_t n_roots, acx_r1cs** out) {
    const HostCircuit& hc = c->hc;
    std::vector<uint64_t> order;
    ACX_TRY(root_order(hc, roots, n_roots, order));
    acx_csr views[3];
    HostCsr P[3];
    for (int k = 0; k < 3; ++k) {
        const HostCsr* src = &c->rows[k];
        if (!order.empty()) { permute_rows(*src, order, P[k]); src = &P[k]; }
        views[k] = acx_csr{src->rowptr.data(), src->col.data(), reinterpret_cast<const acx_fr*>(src->val.data())};
    }
    const acx_csr* mats[3] = {&views[0], &views[1], &views[2]};
    PhaseTimer pt;
    ACX_TRY(r1cs_from_host(ctx, hc.n_rows(), hc.m(), mats, out));
    pt.mark("r1cs_from_host total");
    // the device evaluation plan (generateAssignment on the GPU) is derived on first use: a caller that only verifies
    // never pays for it (28 ms of levelling per 2^20 gates)
    (*out)->plan_src = c;
    c->refs.fetch_add(1);
    (*out)->plan_order = std::move(order);
    return ACX_OK;
}

// Levels, per-gate records and their device copies for acx_r1cs_eval; the caller holds ctx->mu.  Failure is not an error of
// the system: acx_r1cs_eval then reports ACX_ERR_UNSUPPORTED and the host evaluator (acx_circuit_eval) remains.
static void ensure_eval_plan(acx_r1cs* r) {
    if (!r->plan_src) return;
    const acx_circuit* src = r->plan_src;
    r->plan_src = nullptr;
    const HostCircuit& hc = src->hc;
    const std::vector<uint64_t> order = std::move(r->plan_order);
    acx_ctx* ctx = r->ctx;
    PhaseTimer pt;
    struct Release { const acx_circuit* c; ~Release() { circuit_release(c); } } release{src};
    try {
    HostCircuit::EvalPlan plan;
    if (hc.n_gates > 0 && hc.n_gates < 0xffffffffull && hc.build_plan(plan)) {
        pt.mark("  plan: levels");
        const uint64_t ng = hc.n_gates;
        std::vector<uint32_t> inv(hc.n_rows());
        if (order.empty()) for (uint64_t i = 0; i < inv.size(); ++i) inv[i] = (uint32_t)i;
        else for (uint64_t i = 0; i < inv.size(); ++i) inv[order[i]] = (uint32_t)i;
        std::vector<uint32_t> row(ng), wofs(ng + 1), wflat(hc.wires.size());
        uint64_t first_row = 0;
        for (uint64_t g = 0; g < ng; ++g) {
            row[g] = inv[first_row];
            first_row += hc.rows_of_gate(g);
            wofs[g] = (uint32_t)hc.wire_ofs[g];
            if (hc.kind[g] != ACX_GATE_MUL) r->plan_eq_split_inputs.push_back((uint32_t)hc.flat(hc.wires[hc.wire_ofs[g]]));
        }
        wofs[ng] = (uint32_t)hc.wire_ofs[ng];
        for (size_t i = 0; i < hc.wires.size(); ++i) wflat[i] = (uint32_t)hc.flat(hc.wires[i]);
        pt.mark("  plan: gate arrays");
        // level-ordered records of the Mul gates (entry ranges of their A and B rows in the device CSR)
        std::vector<uint32_t> ptr_a(hc.n_rows() + 1), ptr_b(hc.n_rows() + 1);
        // the plan is an optimisation: if anything below fails the system is still valid, only acx_r1cs_eval is not offered
        if (hipMemcpy(ptr_a.data(), r->M[0].ptr, ptr_a.size() * 4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(ptr_b.data(), r->M[1].ptr, ptr_b.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
            return;
        pt.mark("  plan: rowptr download");
        std::vector<uint32_t> mul(plan.items.size() * 4, 0xffffffffu);
        for (size_t t = 0; t < plan.items.size(); ++t) {
            const uint32_t g = plan.items[t];
            if (hc.kind[g] != ACX_GATE_MUL) continue;
            const uint32_t ri = row[g], na = ptr_a[ri + 1] - ptr_a[ri], nb = ptr_b[ri + 1] - ptr_b[ri];
            if (na > 0xffffu || nb > 0xfffeu) continue;          // generic path
            mul[4 * t] = wflat[hc.wire_ofs[g]];
            mul[4 * t + 1] = ptr_a[ri];
            mul[4 * t + 2] = ptr_b[ri];
            mul[4 * t + 3] = na | (nb << 16);
        }
        pt.mark("  plan: mul records");
        auto up = [&](void** dst, const void* src, size_t bytes) -> bool {
            return hipMalloc(dst, bytes ? bytes : 4) == hipSuccess && (bytes == 0 || hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess);
        };
        if (up((void**)&r->ev_items, plan.items.data(), plan.items.size() * 4) && up((void**)&r->ev_row, row.data(), row.size() * 4) &&
            up((void**)&r->ev_wire_ofs, wofs.data(), wofs.size() * 4) && up((void**)&r->ev_wires, wflat.data(), wflat.size() * 4) &&
            up((void**)&r->ev_kind, hc.kind.data(), hc.kind.size()) && up((void**)&r->ev_mul, mul.data(), mul.size() * 4) &&
            up((void**)&r->ev_equal, plan.deferred_equal.data(), plan.deferred_equal.size() * 4) &&
            up((void**)&r->ev_level_ofs, plan.level_ofs.data(), plan.level_ofs.size() * 4) &&
            hipMalloc((void**)&r->ev_cols, plan.items.size() * kEvalLanes * 4 + 4) == hipSuccess) {
            // level-ordered copy of the first four columns of each recorded Mul gate's A and B rows (k_eval_level_lanes)
            const uint64_t lanes = (uint64_t)plan.items.size() * kEvalLanes;
            if (lanes > 0) {
                hipLaunchKernelGGL(k_eval_fill_cols, dim3((unsigned)((lanes + kBlock - 1) / kBlock)), dim3(kBlock), 0, cur_stream(ctx),
                                   (const uint4*)r->ev_mul, (u32)plan.items.size(), (const u32*)r->M[0].idx, (const u32*)r->M[1].idx, r->ev_cols);
                if (hipStreamSynchronize(cur_stream(ctx)) != hipSuccess) return;
            }
            r->has_plan = true;
            r->ev_defer_magic = plan.defer_magic;
            r->n_ev_equal = (uint32_t)plan.deferred_equal.size();
            r->plan_level_ofs = plan.level_ofs;
            r->plan_written = plan.written;
            r->plan_n_in = hc.n_in;
        }
    }
    } catch (const std::bad_alloc&) {
        r->has_plan = false;
    }
    pt.mark("evaluation plan (lazy)");
}

int acx_circuit_to_r1cs(acx_ctx* ctx, const acx_circuit* c, const acx_fr* roots, uint64_t n_roots, acx_r1cs** out) {
    ACX_RANGE();
    if (!ctx || !c || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (c->field != ctx->field) return fail(ACX_ERR_INVALID_ARG, "circuit and context are over different fields");
    return guarded([&]() -> int { return circuit_to_r1cs_impl(ctx, c, roots, n_roots, out); });
}

int acx_circuit_to_r1cs_lists(acx_ctx* ctx, const acx_circuit* c, const acx_fr* roots, const uint32_t* counts, uint64_t n_lists,
                              uint32_t flags, acx_r1cs** out) {
    ACX_RANGE();
    if (!ctx || !c || !out || (n_lists && !counts)) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (c->field != ctx->field) return fail(ACX_ERR_INVALID_ARG, "circuit and context are over different fields");
    if (flags & ~(uint32_t)ACX_ROOTS_REFERENCE_SEMANTICS) return fail(ACX_ERR_INVALID_ARG, "unknown flag");
    return guarded([&]() -> int {
        bool regular = false;
        ACX_TRY(lists_are_regular(c->hc, roots, counts, n_lists, &regular));
        uint64_t total = 0;
        for (uint64_t g = 0; g < n_lists; ++g) total += counts[g];
        if (regular) return circuit_to_r1cs_impl(ctx, c, roots, total, out);      // the ordinary path: rows of the circuit, evaluation plan kept
        if (!(flags & ACX_ROOTS_REFERENCE_SEMANTICS)) {
            ACX_TRY(acx_circuit_check_root_counts(c, counts, n_lists));
            std::vector<uint64_t> order;
            return root_order(c->hc, roots, total, order);                            // reports the duplicate / the bad element
        }
        HostCsr M[3];
        std::vector<H256> distinct;
        std::string msg;
        const int rc = c->hc.build_rows_reference(roots, counts, n_lists, M, distinct, msg);
        if (rc != ACX_OK) return fail(rc, msg);
        acx_csr views[3];
        for (int k = 0; k < 3; ++k) views[k] = acx_csr{M[k].rowptr.data(), M[k].col.data(), reinterpret_cast<const acx_fr*>(M[k].val.data())};
        const acx_csr* mats[3] = {&views[0], &views[1], &views[2]};
        // no evaluation plan: the rows no longer correspond to gates one to one (acx_r1cs_eval reports ACX_ERR_UNSUPPORTED;
        // acx_circuit_eval is the reference's own host fold)
        return r1cs_from_host(ctx, distinct.size(), c->hc.m(), mats, out);
    });
}

// ---------------------------------------------------------------------------------- R1CS
int acx_r1cs_load(acx_ctx* ctx, uint64_t n, uint64_t m, const acx_csr* A, const acx_csr* B, const acx_csr* C,
                  acx_r1cs** out) {
    ACX_RANGE();
    const acx_csr* mats[3] = {A, B, C};
    return guarded([&]() -> int { return r1cs_from_host(ctx, n, m, mats, out); });
}

void acx_r1cs_destroy(acx_r1cs* r) {
    if (!r) return;
    {
        CtxLock lock(r->ctx->mu);
        (void)hipSetDevice(r->ctx->device);
        (void)hipDeviceSynchronize();        // every lane: nothing may still be using this object
        free_r1cs_device(r);
    }
    circuit_release(r->plan_src);
    delete r;
}

int acx_r1cs_dims(const acx_r1cs* r, uint64_t* n, uint64_t* m, uint32_t* log_n, uint64_t nnz[3]) {
    if (!r) return fail(ACX_ERR_INVALID_ARG, "null r1cs");
    if (n) *n = r->n;
    if (m) *m = r->m;
    if (log_n) *log_n = r->log_n;
    if (nnz) for (int k = 0; k < 3; ++k) nnz[k] = r->M[k].nnz;
    return ACX_OK;
}

int acx_r1cs_format(const acx_r1cs* r, uint32_t* small_mask, uint32_t* unit_c, uint64_t* n_long) {
    if (!r) return fail(ACX_ERR_INVALID_ARG, "null r1cs");
    if (small_mask) *small_mask = r->small;
    if (unit_c) *unit_c = r->unit_c ? 1u : 0u;
    if (n_long) *n_long = r->n_long;
    return ACX_OK;
}

int acx_r1cs_export(const acx_r1cs* r, int matrix, uint32_t* rowptr, uint32_t* col, acx_fr* val) {
    if (!r || matrix < 0 || matrix > 2 || !rowptr) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    acx_ctx* c = r->ctx;
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const DevMatrix& M = r->M[matrix];
    HIP_TRY(hipMemcpy(rowptr, M.ptr, (r->n + 1) * 4, hipMemcpyDeviceToHost));
    if (M.nnz && col) HIP_TRY(hipMemcpy(col, M.idx, M.nnz * 4, hipMemcpyDeviceToHost));
    if (M.nnz && val) {
        DevBuf tmp;
        ACX_TRY(tmp.alloc(M.nnz * 32));
        ACX_TRY(download_elements(c, M.val, M.nnz, val, tmp.as<uint4>()));
    }
    return ACX_OK;
}

static int verify_common(acx_r1cs* r, const acx_fr* witness, uint4* d_w, uint64_t* n_bad, uint64_t* first_bad, uint4* d_res,
                         uint4* d_dots, uint64_t dots_stride) {
    acx_ctx* c = r->ctx;
    ACX_TRY(begin_call(c));
    ACX_TRY(upload_elements_async(c, witness, r->m, d_w));
    ACX_TRY(launch_residual(r, d_w, 0, cur_result(c), d_res, d_dots, dots_stride));
    CallSlot& slot = cur_hslot(c);
    ACX_TRY(end_call_fetch(c, &slot));
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));
    if (slot.noncanonical) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    if (n_bad) *n_bad = slot.n_bad;
    if (first_bad) *first_bad = slot.first_bad;
    return ACX_OK;
}

int acx_r1cs_verify(acx_r1cs* r, const acx_fr* witness, int* ok, uint64_t* n_bad, uint64_t* first_bad) {
    ACX_RANGE();
    if (!r || !witness || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
    LaneGuard lane(r->ctx);                                  // concurrent callers overlap: one stream + scratch per lane
    HIP_TRY(hipSetDevice(r->ctx->device));
    uint8_t* base = nullptr;
    ACX_TRY(lane_reserve(r->ctx, r->m * 32, &base));
    uint64_t bad = 0, first = ~0ull;
    ACX_TRY(verify_common(r, witness, (uint4*)base, &bad, &first, nullptr, nullptr, 0));
    *ok = bad == 0;
    if (n_bad) *n_bad = bad;
    if (first_bad) *first_bad = first;
    return ACX_OK;
}

// `all (verifyAssignment qap) assignments` (test/Test/Circuit/Arithmetic.hs:209) in one call: every witness of a chunk
// crosses PCIe in ONE copy, is converted by one kernel, and the chunk is verified by ONE batched launch (blockIdx.y =
// witness; the constraint stream of the system is shared by all of them and stays in L2 / Infinity Cache).
int acx_r1cs_verify_many(acx_r1cs* r, uint64_t count, const acx_fr* witnesses, uint8_t* ok, uint64_t* n_bad, uint64_t* first_bad) {
    ACX_RANGE();
    if (!r || !ok || (count && !witnesses)) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (count == 0) return ACX_OK;
    return guarded([&]() -> int {
        acx_ctx* c = r->ctx;
        LaneGuard lane(c);
        HIP_TRY(hipSetDevice(c->device));
        size_t budget = (size_t)256 << 20;                                 // device bytes of witnesses per chunk
        if (const char* e = std::getenv("ACX_VERIFY_MANY_CHUNK_BYTES")) budget = (size_t)std::max(1ll, std::atoll(e));
        const uint64_t wbytes = r->m * 32;
        const uint64_t chunk_max = std::max<uint64_t>(1, std::min<uint64_t>({count, budget / wbytes, (uint64_t)65535}));
        std::vector<SellSystem> desc(chunk_max);
        std::vector<unsigned long long> res(2 * chunk_max);
        const size_t off_res = align256(chunk_max * wbytes), off_desc = align256(off_res + chunk_max * 16);
        uint8_t* base = nullptr;
        ACX_TRY(lane_reserve(c, off_desc + chunk_max * sizeof(SellSystem), &base));
        StreamDrain drain(cur_stream(c));          // after desc / res: they outlive the copies enqueued from and into them on every exit
        uint4* d_w = (uint4*)base;
        unsigned long long* d_res = (unsigned long long*)(base + off_res);
        SellSystem* d_desc = (SellSystem*)(base + off_desc);
        for (uint64_t done = 0; done < count; done += chunk_max) {
            const uint64_t k = std::min(chunk_max, count - done);
            ACX_TRY(begin_call(c));
            ACX_TRY(upload_elements_async(c, witnesses + done * r->m, k * r->m, d_w));  // canonicity flag fetched below
            for (uint64_t i = 0; i < k; ++i) {
                res[2 * i] = 0; res[2 * i + 1] = ~0ull;
                desc[i] = sell_system(r, d_w + 2 * i * r->m, ResidualOut{d_res + 2 * i, nullptr, nullptr, 0, 0});
            }
            HIP_TRY(hipMemcpyAsync(d_res, res.data(), k * 16, hipMemcpyHostToDevice, cur_stream(c)));
            HIP_TRY(hipMemcpyAsync(d_desc, desc.data(), k * sizeof(SellSystem), hipMemcpyHostToDevice, cur_stream(c)));
            if (r->n_slices) {
                const dim3 grid(sell_grid_x(r->n_slices), (unsigned)k, 1);
                launch_sell(c, sell_spec(r), grid, d_desc, SellSystem{});
                HIP_TRY(hipGetLastError());
            }
            if (r->n_long) ACX_TRY(launch_long_rows(r, nullptr, ResidualOut{}, d_desc, (uint32_t)k));   // one launch per tier for all witnesses
            CallSlot& slot = cur_hslot(c);
            HIP_TRY(hipMemcpyAsync(res.data(), d_res, k * 16, hipMemcpyDeviceToHost, cur_stream(c)));
            ACX_TRY(end_call_fetch(c, &slot));
            HIP_TRY(hipStreamSynchronize(cur_stream(c)));
            if (slot.noncanonical) return fail(ACX_ERR_NONCANONICAL, "element >= p");
            for (uint64_t i = 0; i < k; ++i) {
                ok[done + i] = res[2 * i] == 0;
                if (n_bad) n_bad[done + i] = res[2 * i];
                if (first_bad) first_bad[done + i] = res[2 * i + 1];
            }
        }
        return ACX_OK;
    });
}

int acx_r1cs_eval(acx_r1cs* r, const acx_fr* inputs, const uint8_t* present, uint64_t n_inputs, acx_fr* witness,
                  uint8_t* assigned) {
    ACX_RANGE();
    if (!r || (n_inputs && !inputs)) return fail(ACX_ERR_INVALID_ARG, "null argument");
    acx_ctx* c = r->ctx;
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    ensure_eval_plan(r);
    if (!r->has_plan)
        return fail(ACX_ERR_UNSUPPORTED, "no device evaluation plan (system not built from a single-assignment circuit)");
    // which wires hold a value afterwards (what the QapSet would contain)
    std::vector<uint8_t> as(r->plan_written);
    const uint64_t n_use = std::min<uint64_t>(n_inputs, r->plan_n_in);
    for (uint64_t i = 0; i < n_use; ++i) if (!present || present[i]) as[1 + i] = 1;
    for (uint32_t k : r->plan_eq_split_inputs)
        if (!as[k]) return fail(ACX_ERR_UNDEFINED_WIRE, "evalGate: the impossible happened (Equal/Split input unassigned)");
    // initial witness: constant 1, the given inputs, everything else 0 -- zeroed on the device (the
    // all-zero word is 0 in dev format too); only the head travels over PCIe
    std::vector<acx_fr> w0(1 + n_use);
    std::memset(w0.data(), 0, w0.size() * 32);
    w0[0].b[0] = 1;
    for (uint64_t i = 0; i < n_use; ++i) if (!present || present[i]) w0[1 + i] = inputs[i];
    StreamDrain drain(cur_stream(c));          // after w0: it outlives the copy enqueued from it on every exit
    r->resident_valid = false;
    // no host round trip before the levels: the canonicity flag of the inputs comes back with the call's result slot
    ACX_TRY(begin_call(c));
    HIP_TRY(hipMemsetAsync(r->d_w, 0, r->m * 32, cur_stream(c)));
    ACX_TRY(upload_elements_async(c, w0.data(), w0.size(), r->d_w));
    const CsrDev A{r->M[0].ptr, r->M[0].idx, r->M[0].val}, B{r->M[1].ptr, r->M[1].idx, r->M[1].val};
    const size_t n_levels = r->plan_level_ofs.size() - 1;
    // narrow levels are latency: eight lanes per gate (k_eval_level_lanes); wide ones throughput: a lane per gate
    static const uint32_t lanes_below = [] { const char* e = getenv("ACX_EVAL_LANES_BELOW"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 32768u; }();
    // runs of narrow levels (<= kEvalFusedGates gates each) go to ONE workgroup in ONE launch: a level costs a barrier there
    static const bool fuse = [] { const char* e = getenv("ACX_EVAL_FUSED"); return !e || strcmp(e, "0") != 0; }();
    auto width = [&](size_t l) { return r->plan_level_ofs[l + 1] - r->plan_level_ofs[l]; };
    const uint32_t dm = r->ev_defer_magic ? 1u : 0u;
    for (size_t l = 0; l < n_levels;) {
        const uint32_t lo = r->plan_level_ofs[l], cnt = width(l);
        if (fuse && cnt <= kEvalFusedGates) {
            size_t e = l + 1;
            while (e < n_levels && width(e) <= kEvalFusedGates) ++e;
            if (e - l >= 2) {
                const EvalGates G{r->ev_items, 0u, r->ev_kind, r->ev_row, r->ev_wire_ofs, r->ev_wires, r->ev_mul, r->ev_cols, dm};
                DISPATCH_FIELD(c, hipLaunchKernelGGL((k_eval_levels_fused<F>), dim3(1), dim3(kEvalFusedBlock), 0, cur_stream(c),
                                                     G, (const u32*)r->ev_level_ofs, (u32)l, (u32)e, A, B, r->d_w));
                l = e;
                continue;
            }
        }
        ++l;
        if (cnt == 0) continue;
        const EvalGates G{r->ev_items + lo, cnt, r->ev_kind, r->ev_row, r->ev_wire_ofs, r->ev_wires, r->ev_mul + lo, r->ev_cols + (size_t)lo * kEvalLanes, dm};
        if (cnt < lanes_below) {
            const uint32_t per_block = kBlock / kEvalLanes;
            DISPATCH_FIELD(c, hipLaunchKernelGGL((k_eval_level_lanes<F>), dim3((cnt + per_block - 1) / per_block), dim3(kBlock), 0, cur_stream(c),
                                                 G, A, B, r->d_w));
        } else {
            DISPATCH_FIELD(c, hipLaunchKernelGGL((k_eval_level<F>), dim3((cnt + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(c),
                                                 G, A, B, r->d_w));
        }
    }
    if (r->n_ev_equal)
        DISPATCH_FIELD(c, hipLaunchKernelGGL((k_eval_magic<F>), dim3((r->n_ev_equal + kSlice - 1) / kSlice), dim3(kSlice), 0, cur_stream(c),
                                             (const u32*)r->ev_equal, r->n_ev_equal, (const u32*)r->ev_wire_ofs, (const u32*)r->ev_wires, r->d_w));
    HIP_TRY(hipGetLastError());
    CallSlot& slot = cur_hslot(c);
    if (witness) {
        if (!r->d_w_canon) HIP_TRY(hipMalloc((void**)&r->d_w_canon, r->m * 32));
        ACX_TRY(launch_convert(c, false, r->d_w, r->d_w_canon, r->m, nullptr));
        HIP_TRY(hipMemcpyAsync(witness, r->d_w_canon, r->m * 32, hipMemcpyDeviceToHost, cur_stream(c)));
    }
    ACX_TRY(end_call_fetch(c, &slot));
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));
    if (slot.noncanonical) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    r->resident_valid = true;
    if (assigned) std::memcpy(assigned, as.data(), as.size());
    return ACX_OK;
}

int acx_r1cs_verify_resident(acx_r1cs* r, int* ok, uint64_t* n_bad, uint64_t* first_bad) {
    ACX_RANGE();
    if (!r || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
    acx_ctx* c = r->ctx;
    CtxLock lock(c->mu);
    if (!r->resident_valid)
        return fail(ACX_ERR_UNSUPPORTED, "no resident witness: acx_r1cs_eval has not run on this system (or acx_naive_h has used the buffer since)");
    HIP_TRY(hipSetDevice(c->device));
    const unsigned long long init[2] = {0ull, ~0ull};
    HIP_TRY(hipMemcpyAsync(cur_result(c), init, 16, hipMemcpyHostToDevice, cur_stream(c)));
    ACX_TRY(launch_residual(r, r->d_w, 0, cur_result(c), nullptr, nullptr, 0));
    unsigned long long res[2];
    HIP_TRY(hipMemcpyAsync(res, cur_result(c), 16, hipMemcpyDeviceToHost, cur_stream(c)));
    HIP_TRY(hipStreamSynchronize(cur_stream(c)));
    *ok = res[0] == 0;
    if (n_bad) *n_bad = res[0];
    if (first_bad) *first_bad = res[1];
    return ACX_OK;
}

int acx_r1cs_residuals(acx_r1cs* r, const acx_fr* witness, acx_fr* out) {
    ACX_RANGE();
    if (!r || !witness || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    acx_ctx* c = r->ctx;
    LaneGuard lane(c);
    HIP_TRY(hipSetDevice(c->device));
    uint8_t* base = nullptr;
    const size_t wb = align256(r->m * 32);
    ACX_TRY(lane_reserve(c, wb + r->n * 32, &base));
    uint4* res = (uint4*)(base + wb);
    ACX_TRY(verify_common(r, witness, (uint4*)base, nullptr, nullptr, res, nullptr, 0));
    return download_elements(c, res, r->n, out, res);
}

// verificationWitnessZk on device-resident data (caller holds ctx->mu): residual dots -> 3 iNTT -> 2 coset NTT (L, R) ->
// pointwise -> coset iNTT -> minus O0 / z in coefficient form (+ the zero-knowledge terms).  d_h receives N+1 dev elements, not stripped.
static int get_h_scale(acx_ctx* c, uint32_t log_n, const H256& g, const uint4** out);
static int qap_h_dev_locked(acx_r1cs* r, const uint4* d_w, const H256* dl, uint4* d_h, unsigned long long* d_result,
                            uint4* d /* 5N elements of scratch */) {
    acx_ctx* c = r->ctx;
    const HostField& hf = c->hf;
    if ((int)r->log_n + 1 > hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "coset needs log_n + 1 <= two-adicity");
    const uint64_t N = 1ull << r->log_n;
    uint4* keep = d + 6 * N;                                          // dots (3N) + kept L0, R0 (2N)
    if (N > r->n)                                                    // rows n..N-1 are the zero padding
        for (int k = 0; k < 3; ++k) HIP_TRY(hipMemsetAsync(d + 2 * ((uint64_t)k * N + r->n), 0, (N - r->n) * 32, cur_stream(c)));
    const bool zk = dl && !(dl[0].is_zero() && dl[1].is_zero() && dl[2].is_zero());
    // coset: shift = multiplicative generator g (g^N != 1); z = g^N - 1 is the target polynomial on it
    const H256 g = hf.generator();
    const H256 zinv = hf.inv(hf.sub(hf.pow_u64(g, N), hf.one()));
    const H256 mzinv = hf.sub(hf.zero(), zinv);
    // Without the zero-knowledge terms 1/z and -1/z ride on the stored dot products (one product per row in a launch that waits
    // for memory): (L/z) R - O/z is then what the rest of the pipeline forms, with no pass over the product and no scaled
    // subtraction.  The two constants live beside the system (they depend on N alone: r1cs_from_host).
    // (a system loaded while log_n + 1 exceeded the two-adicity has no pair of its own; acx_ctx_set_root may have raised the
    // two-adicity since: the context's cache supplies the pair then -- the rest of the pipeline multiplies by one and RELIES on it)
    const uint4* hscale = r->d_hscale;
    if (!zk && !hscale) ACX_TRY(get_h_scale(c, r->log_n, g, &hscale));
    ACX_TRY(launch_residual(r, d_w, 0, d_result, nullptr, d, N, 0, 0, zk ? nullptr : hscale));
    // evaluations on <omega> -> coefficients of L0, R0, O0; L0 and R0 -> evaluations on g<omega>.  O0 stays in coefficient
    // form: h = icoset((L R - O)/z) = icoset(L R / z) - O0 / z by linearity (icoset after coset is the identity), which
    // drops one of the seven transforms.  Without the zero-knowledge terms nobody needs the plain coefficients of L0 and
    // R0, so their factor g^i rides on the inverse transform's closing multiplication.
    uint4* O0 = d + 4 * N;
    // fused: all three inverse transforms in one batched launch, g^i riding on L and R.  Plans of two or more passes leave O
    // plain (its closing step is the multiplication-free reduction); single-pass sizes put g^i on O too and the closing
    // subtraction takes it off again from the two-level table (k_axpy_geo)
    bool o_plain = false;           // O came out of the batched launch WITHOUT the coset factor (plans of two or more passes)
    int fused = zk ? ACX_ERR_UNSUPPORTED : ntt_dev_locked(c, d, r->log_n, 3, 1, nullptr, &g, 2, &o_plain);
    if (fused == ACX_OK) {
        ACX_TRY(ntt_dev_locked(c, d, r->log_n, 2, 0, nullptr));
    } else {
        ACX_TRY(ntt_dev_locked(c, d, r->log_n, 3, 1, nullptr));
        if (zk) HIP_TRY(hipMemcpyAsync(keep, d, 2 * N * 32, hipMemcpyDeviceToDevice, cur_stream(c)));
        ACX_TRY(ntt_dev_locked(c, d, r->log_n, 2, 0, &g));
    }
    if (zk) {
        DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pointwise_h<F>), dim3(grid_for(c, N)), dim3(kBlock), 0, cur_stream(c), (const uint4*)d,
                                             (const uint4*)(d + 2 * N), (const uint4*)nullptr, d_h, N, dev_arg(hf, zinv), 0u));
        ACX_TRY(ntt_dev_locked(c, d_h, r->log_n, 1, 1, &g));
        // (L0+d1 T)(R0+d2 T) - (O0+d3 T) = T * (h0 + d1 R0 + d2 L0 + d1 d2 T - d3),  T = x^N - 1
        const uint4* L0 = keep;
        const uint4* R0 = L0 + 2 * N;
        const H256 d12 = hf.mul(dl[0], dl[1]);
        DISPATCH_FIELD(c, {
            hipLaunchKernelGGL((k_axpy3<F>), dim3(grid_for(c, N)), dim3(kBlock), 0, cur_stream(c), d_h, R0, L0, (const uint4*)O0, N,
                               dev_arg(hf, dl[0]), dev_arg(hf, dl[1]), dev_arg(hf, mzinv));
            hipLaunchKernelGGL((k_h_fix<F>), dim3(1), dim3(64), 0, cur_stream(c), d_h, N, dev_arg(hf, hf.add(d12, dl[2])), dev_arg(hf, d12));
        });
        HIP_TRY(hipGetLastError());
        return ACX_OK;
    }
    // d = (L/z) on the coset, d + N = R on the coset, O0 = -O/z in coefficient form (times g^i when !o_plain of a fused launch).
    // The last transform takes the product as its first pass loads the points and adds O0 behind its closing step -- where
    // the plan of this size can (two or more passes of k_ntt_r4); otherwise the two passes over the vectors run as kernels.
    const bool o_has_g = fused == ACX_OK && !o_plain;
    const H256 one = hf.one();
    const int last = ntt_dev_locked(c, d_h, r->log_n, 1, 1, &g, nullptr, 0, nullptr, d, d + 2 * N, o_has_g ? nullptr : O0);
    bool o_added = last == ACX_OK && !o_has_g;
    if (last == ACX_OK) {
        HIP_TRY(hipMemsetAsync(d_h + 2 * N, 0, 32, cur_stream(c)));                          // h has N + 1 coefficients
    } else if (last == ACX_ERR_UNSUPPORTED) {
        DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pointwise_h<F>), dim3(grid_for(c, N)), dim3(kBlock), 0, cur_stream(c), (const uint4*)d,
                                             (const uint4*)(d + 2 * N), (const uint4*)nullptr, d_h, N, dev_arg(hf, one), 1u));
        ACX_TRY(ntt_dev_locked(c, d_h, r->log_n, 1, 1, &g));
    } else {
        return last;
    }
    if (o_has_g) {
        uint4 *glo = nullptr, *ghi = nullptr;
        ACX_TRY(get_coset_tables(c, hf.inv(g), r->log_n, 0, &glo, &ghi, 0));
        DISPATCH_FIELD(c, hipLaunchKernelGGL((k_axpy_geo<F>), dim3(grid_for(c, N)), dim3(kBlock), 0, cur_stream(c), d_h, (const uint4*)O0, N,
                                             (const uint4*)glo, (const uint4*)ghi, dev_arg(hf, one)));
    } else if (!o_added) {
        DISPATCH_FIELD(c, hipLaunchKernelGGL((k_axpy3<F>), dim3(grid_for(c, N)), dim3(kBlock), 0, cur_stream(c), d_h, (const uint4*)nullptr,
                                             (const uint4*)nullptr, (const uint4*)O0, N, dev_arg(hf, one), dev_arg(hf, one), dev_arg(hf, one)));
    }
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_qap_h_dev(acx_r1cs* r, const void* d_witness, const acx_fr* delta, void* d_h, uint64_t* d_result) {
    ACX_RANGE();
    if (!r || !d_witness || !d_h || !d_result) return fail(ACX_ERR_INVALID_ARG, "null argument");
    acx_ctx* c = r->ctx;
    H256 dl[3];
    if (delta) for (int k = 0; k < 3; ++k) ACX_TRY(read_h256(&delta[k], c->hf, dl[k]));
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t N = 1ull << r->log_n;
    if (!r->qh) HIP_TRY(hipMalloc((void**)&r->qh, 5 * N * 32));      // scratch of the device-pointer path: lives with the system
    return qap_h_dev_locked(r, (const uint4*)d_witness, delta ? dl : nullptr, (uint4*)d_h, (unsigned long long*)d_result, r->qh);
}

int acx_qap_h(acx_r1cs* r, const acx_fr* witness, const acx_fr* delta, acx_fr* out_h, uint64_t* h_len, int* ok) {
    ACX_RANGE();
    if (!r || !witness || !out_h || !h_len || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
    acx_ctx* c = r->ctx;
    const HostField& hf = c->hf;
    if ((int)r->log_n + 1 > hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "coset needs log_n + 1 <= two-adicity");
    H256 dl[3];
    if (delta) for (int k = 0; k < 3; ++k) ACX_TRY(read_h256(&delta[k], hf, dl[k]));
    LaneGuard lane(c);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t N = 1ull << r->log_n;
    uint8_t* base = nullptr;
    const size_t wb = align256(r->m * 32), hb = align256(2 * (N + 1) * 32);      // witness | h (N+1) + conversion scratch | 5N pipeline scratch
    ACX_TRY(lane_reserve(c, wb + hb + 5 * N * 32, &base));
    uint4* d_wit = (uint4*)base;
    uint4* d_h = (uint4*)(base + wb);
    ACX_TRY(begin_call(c));
    ACX_TRY(upload_elements_async(c, witness, r->m, d_wit));
    ACX_TRY(qap_h_dev_locked(r, d_wit, delta ? dl : nullptr, d_h, cur_result(c), (uint4*)(base + wb + hb)));
    CallSlot& slot = cur_hslot(c);
    ACX_TRY(end_call_fetch(c, &slot));
    ACX_TRY(download_elements(c, d_h, N + 1, out_h, d_h + 2 * (N + 1)));                 // synchronises the stream
    if (slot.noncanonical) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    *ok = slot.n_bad == 0;
    uint64_t len = N + 1;
    static const uint8_t zero32[32] = {0};
    while (len > 0 && std::memcmp(out_h[len - 1].b, zero32, 32) == 0) --len;
    *h_len = len;
    return ACX_OK;
}

int acx_qap_columns_dev(acx_r1cs* r, int matrix, uint64_t wire_begin, uint64_t wire_count, void* d_out, uint64_t* d_len) {
    ACX_RANGE();
    if (!r || matrix < 0 || matrix > 2 || !d_out) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    if (wire_begin + wire_count > r->m) return fail(ACX_ERR_INVALID_ARG, "wire range exceeds m");
    acx_ctx* c = r->ctx;
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    ACX_TRY(ensure_csc(r));
    return qap_columns_core(r, matrix, wire_begin, wire_count, (uint4*)d_out, (unsigned long long*)d_len);
}

int acx_qap_columns(acx_r1cs* r, int matrix, uint64_t wire_begin, uint64_t wire_count, acx_fr* out,
                    uint64_t* out_len) {
    return qap_columns_host(r, matrix, wire_begin, wire_count, out, out_len, 1ull << 30);
}

}  // extern "C"

// acx_qap_columns with the size of a device-side batch of coefficients bounded by the caller (the N-GPU handle runs one of
// these per shard and bounds the sum)
int qap_columns_host(acx_r1cs* r, int matrix, uint64_t wire_begin, uint64_t wire_count, acx_fr* out, uint64_t* out_len,
                     uint64_t max_batch_bytes) {
    AbiRange acx_range_("acx_qap_columns");
    if (!r || matrix < 0 || matrix > 2 || !out) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    if (wire_begin + wire_count > r->m) return fail(ACX_ERR_INVALID_ARG, "wire range exceeds m");
    if (wire_count == 0) return ACX_OK;
    acx_ctx* c = r->ctx;
    LaneGuard lane(c);
    HIP_TRY(hipSetDevice(c->device));
    ACX_TRY(ensure_csc(r));
    const uint64_t N = 1ull << r->log_n;
    // Wire batches of bounded size (<= 1 GiB of coefficients each), double buffered: while the host thread sits in
    // the blocking device-to-host copy of batch k (on the lane's copy stream), batch k+1 is already scattered and
    // transformed on the lane's compute stream.
    const uint64_t chunk = std::min<uint64_t>(std::max<uint64_t>(1, max_batch_bytes / (N * 32)), wire_count);
    const size_t cb = align256(chunk * N * 32), lb = align256(chunk * 8);
    uint8_t* base = nullptr;
    ACX_TRY(lane_reserve(c, 2 * (cb + lb), &base));
    acx_ctx::Lane* ln = t_lane;
    if (!ln->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ln->copy_stream, hipStreamNonBlocking));
    if (!ln->ev[0]) for (auto& e : ln->ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    auto buf = [&](uint64_t k) { return (uint4*)(base + (k & 1) * (cb + lb)); };
    auto lens = [&](uint64_t k) { return (unsigned long long*)(base + (k & 1) * (cb + lb) + cb); };
    auto fetch = [&](uint64_t k) -> int {          // batch k: wait for its kernels, copy coefficients (+ lengths) out
        const uint64_t w0 = k * chunk, cnt = std::min(chunk, wire_count - w0);
        HIP_TRY(hipStreamWaitEvent(ln->copy_stream, ln->ev[k & 1], 0));
        HIP_TRY(hipMemcpyAsync(out + w0 * N, buf(k), cnt * N * 32, hipMemcpyDeviceToHost, ln->copy_stream));
        if (out_len) HIP_TRY(hipMemcpyAsync(out_len + w0, lens(k), cnt * 8, hipMemcpyDeviceToHost, ln->copy_stream));
        HIP_TRY(hipStreamSynchronize(ln->copy_stream));
        return ACX_OK;
    };
    const uint64_t n_chunks = (wire_count + chunk - 1) / chunk;
    for (uint64_t k = 0; k < n_chunks; ++k) {
        const uint64_t w0 = k * chunk, cnt = std::min(chunk, wire_count - w0);
        ACX_TRY(qap_columns_core(r, matrix, wire_begin + w0, cnt, buf(k), lens(k)));
        ACX_TRY(launch_convert(c, false, buf(k), buf(k), cnt * N, nullptr));     // dev -> canonical in place
        HIP_TRY(hipEventRecord(ln->ev[k & 1], cur_stream(c)));
        if (k > 0) ACX_TRY(fetch(k - 1));          // blocks the host; the GPU works on batch k meanwhile
    }
    return fetch(n_chunks - 1);
}

// Scratch of the host-buffer entry points (lane arenas, transform ping-pong buffers) back to the device: what a caller that
// keeps many contexts on one device (the N-GPU handle with a repeated ordinal) does after a call with large outputs.
void ctx_trim_scratch(acx_ctx* c) {
    (void)hipSetDevice(c->device);
    for (auto& ln : c->lanes) {
        // a lane in use keeps its scratch (and the lane mutex is the ONLY lock taken here: a call on a lane takes ctx->mu while
        // holding its lane, so taking them in the other order could deadlock against it)
        std::unique_lock<std::mutex> g(ln.mu, std::try_to_lock);
        if (!g.owns_lock()) continue;
        if (ln.stream) (void)hipStreamSynchronize(ln.stream);
        if (ln.copy_stream) (void)hipStreamSynchronize(ln.copy_stream);
        if (ln.arena) { (void)hipFree(ln.arena); ln.arena = nullptr; ln.arena_bytes = 0; }
        if (ln.ntt_scratch) { (void)hipFree(ln.ntt_scratch); ln.ntt_scratch = nullptr; ln.ntt_scratch_bytes = 0; }
    }
}

// A constraint system of which ONE DEVICE holds only the column view of some wires (the N-GPU handle's share of
// `createPolynomialsFFT`, src/QAP.hs:512-525: a wire's interpolation needs every row of ITS column and nothing else): an
// acx_r1cs with no row form at all -- m = the number of local wires, T[k] = the CSC of matrix k over them (local column
// numbers, canonical values in, dev format on the device).  Serves acx_qap_columns / qap_columns_host only.
struct HostCsc {
    std::vector<uint32_t> colptr, rowidx, colid;
    std::vector<acx_fr> val;
};
int r1cs_column_slice_from_host(acx_ctx* ctx, uint64_t n, uint32_t log_n, uint64_t m_local, const HostCsc csc[3], acx_r1cs** out) {
    HIP_TRY(hipSetDevice(ctx->device));
    std::unique_ptr<acx_r1cs> r(new acx_r1cs());
    r->ctx = ctx; r->n = n; r->m = m_local; r->log_n = log_n;
    int rc = ACX_OK;
    {
        LaneGuard lane(ctx);
        auto build = [&]() -> int {
            for (int k = 0; k < 3; ++k) {
                DevMatrix& T = r->T[k];
                const uint64_t nnz = csc[k].rowidx.size();
                T.nnz = nnz;
                HIP_TRY(hipMalloc((void**)&T.ptr, (m_local + 1) * 4));
                HIP_TRY(hipMalloc((void**)&T.idx, std::max<uint64_t>(nnz, 1) * 4));
                HIP_TRY(hipMalloc((void**)&T.colid, std::max<uint64_t>(nnz, 1) * 4));
                HIP_TRY(hipMalloc((void**)&T.val, std::max<uint64_t>(nnz, 1) * 32));
                HIP_TRY(hipMemcpyAsync(T.ptr, csc[k].colptr.data(), (m_local + 1) * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
                if (nnz) {
                    HIP_TRY(hipMemcpyAsync(T.idx, csc[k].rowidx.data(), nnz * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
                    HIP_TRY(hipMemcpyAsync(T.colid, csc[k].colid.data(), nnz * 4, hipMemcpyHostToDevice, cur_stream(ctx)));
                    ACX_TRY(upload_elements(ctx, csc[k].val.data(), nnz, T.val));      // canonical -> dev, canonicity checked, synchronises
                }
                T.h_ptr = csc[k].colptr;
            }
            HIP_TRY(hipStreamSynchronize(cur_stream(ctx)));
            return ACX_OK;
        };
        rc = build();
    }
    if (rc != ACX_OK) {
        (void)hipDeviceSynchronize();
        free_r1cs_device(r.get());
        return rc;
    }
    r->has_csc = true;
    *out = r.release();
    return ACX_OK;
}

extern "C" {

// ---------------------------------------------------------------------------------- NTT
int acx_ntt(acx_ctx* c, uint32_t log_n, uint64_t batch, int inverse, const acx_fr* shift, const acx_fr* in,
            acx_fr* out) {
    ACX_RANGE();
    if (!c || !in || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if ((int)log_n > c->hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "log_n exceeds the field's two-adicity");
    LaneGuard lane(c);
    HIP_TRY(hipSetDevice(c->device));
    H256 sh;
    if (shift) {
        ACX_TRY(read_h256(shift, c->hf, sh));
        if (sh.is_zero()) return fail(ACX_ERR_INVALID_ARG, "coset shift must be nonzero");
    }
    const uint64_t total = batch << log_n;
    if (total == 0) return ACX_OK;
    uint8_t* base = nullptr;
    ACX_TRY(lane_reserve(c, 2 * total * 32, &base));
    uint4* buf = (uint4*)base;
    ACX_TRY(upload_elements(c, in, total, buf));
    ACX_TRY(ntt_dev_locked(c, buf, log_n, batch, inverse, shift ? &sh : nullptr));
    return download_elements(c, buf, total, out, buf + 2 * total);
}

// ---------------------------------------------------------------------------------- device API
int acx_dev_from_canonical(acx_ctx* c, uint64_t count, const void* d_in, void* d_out, uint32_t* d_err) {
    if (!c || !d_in || !d_out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    return launch_convert(c, true, d_in, d_out, count, d_err);
}

int acx_dev_to_canonical(acx_ctx* c, uint64_t count, const void* d_in, void* d_out) {
    if (!c || !d_in || !d_out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    return launch_convert(c, false, d_in, d_out, count, nullptr);
}

int acx_r1cs_verify_dev(acx_r1cs* r, const void* d_witness, uint64_t row_offset, uint64_t* d_result,
                        void* d_residuals, void* d_dots) {
    ACX_RANGE();
    if (!r || !d_witness || !d_result) return fail(ACX_ERR_INVALID_ARG, "null argument");
    CtxLock lock(r->ctx->mu);
    HIP_TRY(hipSetDevice(r->ctx->device));
    return launch_residual(r, (const uint4*)d_witness, row_offset, (unsigned long long*)d_result,
                           (uint4*)d_residuals, (uint4*)d_dots, 1ull << r->log_n);
}

// {1/z, -1/z}, z = g^N - 1, for N = 2^log_n and the coset shift g (Montgomery) as two dev elements, cached per context
// (caller holds c->mu): what the h(x) pipeline lets ride on the stored dot products (qap_h_dev_locked).  A rank of a
// distributed job needs them for the GLOBAL N, which its own system (N / world rows) does not know.
static int get_h_scale(acx_ctx* c, uint32_t log_n, const H256& g, const uint4** out) {
    const HostField& hf = c->hf;
    if ((int)log_n + 1 > hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "coset needs log_n + 1 <= two-adicity");
    const std::pair<uint32_t, std::array<uint64_t, 4>> key{log_n, {g.l[0], g.l[1], g.l[2], g.l[3]}};
    auto it = c->h_scale.find(key);
    if (it != c->h_scale.end()) { *out = it->second; return ACX_OK; }
    const H256 z = hf.sub(hf.pow_u64(g, 1ull << log_n), hf.one());
    if (z.is_zero()) return fail(ACX_ERR_INVALID_ARG, "coset shift lies in the transform's own subgroup (shift^N = 1)");
    const H256 zinv = hf.inv(z);
    const H256 pair[2] = {hf.to_dev_word(zinv), hf.to_dev_word(hf.sub(hf.zero(), zinv))};
    uint4* d = nullptr;
    HIP_TRY(hipMalloc((void**)&d, 64));
    const hipError_t e = hipMemcpy(d, pair, 64, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d); HIP_TRY(e); }
    c->h_scale[key] = d;
    *out = d;
    return ACX_OK;
}

int acx_r1cs_dots_h_dev(acx_r1cs* r, const void* d_witness, uint64_t row_offset, uint64_t* d_result, void* d_dots, uint32_t h_log_n,
                        const acx_fr* shift) {
    ACX_RANGE();
    if (!r || !d_witness || !d_result || !d_dots) return fail(ACX_ERR_INVALID_ARG, "null argument");
    CtxLock lock(r->ctx->mu);
    HIP_TRY(hipSetDevice(r->ctx->device));
    H256 g = r->ctx->hf.generator();
    if (shift) {
        ACX_TRY(read_h256(shift, r->ctx->hf, g));
        if (g.is_zero()) return fail(ACX_ERR_INVALID_ARG, "coset shift must be nonzero");
    }
    const uint4* scale = nullptr;
    ACX_TRY(get_h_scale(r->ctx, h_log_n, g, &scale));
    return launch_residual(r, (const uint4*)d_witness, row_offset, (unsigned long long*)d_result, nullptr, (uint4*)d_dots,
                           1ull << r->log_n, 0, 0, scale);
}

int acx_qap_pointwise_dev(acx_ctx* c, uint32_t log_n, uint64_t count, const acx_fr* shift, const void* d_a, const void* d_b,
                          const void* d_c, void* d_out) {
    if (!c || !shift || !d_a || !d_b || !d_out) return fail(ACX_ERR_INVALID_ARG, "null argument");      // d_c may be NULL
    if (count == 0) return ACX_OK;
    const HostField& hf = c->hf;
    H256 g;
    ACX_TRY(read_h256(shift, hf, g));
    const H256 z = hf.sub(hf.pow_u64(g, 1ull << log_n), hf.one());
    if (z.is_zero()) return fail(ACX_ERR_INVALID_ARG, "shift^N = 1: the coset meets the evaluation domain");
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_pointwise_h<F>), dim3(grid_for(c, count)), dim3(kBlock), 0, cur_stream(c), (const uint4*)d_a,
                                         (const uint4*)d_b, (const uint4*)d_c, (uint4*)d_out, count, dev_arg(hf, hf.inv(z)), 0u));
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_qap_sub_o_dev(acx_ctx* c, uint32_t log_n, uint64_t count, const acx_fr* shift, void* d_h, const void* d_o) {
    if (!c || !shift || !d_h || !d_o) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (count == 0) return ACX_OK;
    const HostField& hf = c->hf;
    H256 g;
    ACX_TRY(read_h256(shift, hf, g));
    const H256 z = hf.sub(hf.pow_u64(g, 1ull << log_n), hf.one());
    if (z.is_zero()) return fail(ACX_ERR_INVALID_ARG, "shift^N = 1: the coset meets the evaluation domain");
    const H256 mzinv = hf.sub(hf.zero(), hf.inv(z));
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_axpy3<F>), dim3(grid_for(c, count)), dim3(kBlock), 0, cur_stream(c), (uint4*)d_h, (const uint4*)nullptr,
                                         (const uint4*)nullptr, (const uint4*)d_o, count, dev_arg(hf, mzinv), dev_arg(hf, mzinv), dev_arg(hf, mzinv)));
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int acx_ntt_dist_step_ex_dev(acx_ctx* c, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int inverse, int step,
                             uint32_t flags, const acx_fr* shift, const void* d_in, void* d_out) {
    return acx_ntt_dist_step_fused_dev(c, log_n, log_r, world, rank, inverse, step, flags, shift, d_in, nullptr, nullptr, d_out);
}

int acx_ntt_dist_step_fused_dev(acx_ctx* c, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int inverse, int step,
                                uint32_t flags, const acx_fr* shift, const void* d_in, const void* d_mul, const void* d_add, void* d_out) {
    ACX_RANGE();
    if (!c || !d_in || !d_out || (step != 0 && step != 1)) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    if (d_mul && !inverse && step == 0 && shift) return fail(ACX_ERR_UNSUPPORTED, "no product on load of a forward coset step");
    if (d_add == d_out || d_mul == d_out) return fail(ACX_ERR_INVALID_ARG, "steps are out of place");
    if (flags & ~(uint32_t)ACX_DIST_ROWS_T) return fail(ACX_ERR_INVALID_ARG, "unknown flag");
    if ((flags & ACX_DIST_ROWS_T) && !(inverse && step == 0)) return fail(ACX_ERR_INVALID_ARG, "ACX_DIST_ROWS_T applies to inverse step 0");
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    H256 sh;
    if (shift) {
        ACX_TRY(read_h256(shift, c->hf, sh));
        if (sh.is_zero()) return fail(ACX_ERR_INVALID_ARG, "coset shift must be nonzero");
    }
    return ntt_dist_step_locked(c, log_n, log_r, world, rank, inverse, step, shift ? &sh : nullptr, (const uint4*)d_in,
                                (uint4*)d_out, (flags & ACX_DIST_ROWS_T) != 0, (const uint4*)d_mul, (const uint4*)d_add);
}

int acx_ntt_dist_step_dev(acx_ctx* c, uint32_t log_n, uint32_t log_r, uint32_t world, uint32_t rank, int inverse, int step,
                          const acx_fr* shift, const void* d_in, void* d_out) {
    return acx_ntt_dist_step_ex_dev(c, log_n, log_r, world, rank, inverse, step, 0, shift, d_in, d_out);
}

// ---------------------------------------------------------------------------------- naive-roots path
void acx_naive_destroy(acx_naive* nv) {
    if (!nv) return;
    {
        CtxLock lock(nv->r->ctx->mu);
        (void)hipSetDevice(nv->r->ctx->device);
        (void)hipDeviceSynchronize();        // every lane: nothing may still be using this object
        if (nv->roots) (void)hipFree(nv->roots);
        if (nv->tcoef) (void)hipFree(nv->tcoef);
        if (nv->winv) (void)hipFree(nv->winv);
        if (nv->Q) (void)hipFree(nv->Q);
    }
    delete nv;
}

int acx_naive_create(acx_r1cs* r, const acx_fr* roots, uint64_t n_roots, acx_naive** out) {
    ACX_RANGE();
    if (!r || !roots || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (n_roots != r->n) return fail(ACX_ERR_ROOT_COUNT, "one root per constraint row is required");
    if (r->n == 0 || r->n > 4096) return fail(ACX_ERR_TOO_LARGE, "naive interpolation supports 1..4096 rows");
    acx_ctx* c = r->ctx;
    const HostField& hf = c->hf;
    for (uint64_t i = 0; i < n_roots; ++i) {
        H256 a, b;
        std::memcpy(a.l, roots[i].b, 32);
        if (!hf.is_canonical(a)) return fail(ACX_ERR_NONCANONICAL, "root >= p");
        if (i) {
            std::memcpy(b.l, roots[i - 1].b, 32);
            if (h256_cmp(b, a) >= 0) return fail(ACX_ERR_DUPLICATE_ROOT, "roots must be distinct and ascending (row order)");
        }
    }
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    acx_naive* nv = new (std::nothrow) acx_naive();
    if (!nv) return fail(ACX_ERR_OOM, "host allocation failed");
    nv->r = r;
    const uint32_t n = (uint32_t)r->n;
    nv->n = n;
    DevBuf tmp;
    int rc = tmp.alloc((size_t)(n + 1) * 32);
    auto bail = [&](int code) { if (nv->roots) (void)hipFree(nv->roots); if (nv->tcoef) (void)hipFree(nv->tcoef);
                                if (nv->winv) (void)hipFree(nv->winv); if (nv->Q) (void)hipFree(nv->Q); delete nv; return code; };
    if (rc != ACX_OK) return bail(rc);
    if (hipMalloc((void**)&nv->roots, (size_t)n * 32) != hipSuccess || hipMalloc((void**)&nv->tcoef, (size_t)(n + 1) * 32) != hipSuccess ||
        hipMalloc((void**)&nv->winv, (size_t)n * 32) != hipSuccess || hipMalloc((void**)&nv->Q, (size_t)n * n * 32) != hipSuccess)
        return bail(fail(ACX_ERR_OOM, "device allocation failed"));
    rc = upload_elements(c, roots, n, nv->roots);
    if (rc != ACX_OK) return bail(rc);
    Exp256 pm2;
    {
        H256 e = hf.modulus();
        e.l[0] -= 2;   // p is odd and > 2: no borrow
        for (int i = 0; i < 8; ++i) pm2.w[i] = (u32)(e.l[i / 2] >> (32 * (i % 2)));
    }
    DISPATCH_FIELD(c, {
        hipLaunchKernelGGL((k_poly_from_roots<F>), dim3(1), dim3(1024), 0, cur_stream(c), (const uint4*)nv->roots, n, nv->tcoef, tmp.as<uint4>());
        hipLaunchKernelGGL((k_bary_inv<F>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(c), (const uint4*)nv->roots, n, nv->winv, pm2);
        hipLaunchKernelGGL((k_build_q<F>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(c), (const uint4*)nv->roots,
                           (const uint4*)nv->tcoef, (const uint4*)nv->winv, n, nv->Q);
    });
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(cur_stream(c)) != hipSuccess) return bail(fail(ACX_ERR_HIP, "naive setup kernels failed"));
    *out = nv;
    return ACX_OK;
}

int acx_naive_target(acx_naive* nv, acx_fr* out) {
    if (!nv || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    acx_ctx* c = nv->r->ctx;
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    DevBuf tmp;
    ACX_TRY(tmp.alloc((size_t)(nv->n + 1) * 32));
    return download_elements(c, nv->tcoef, nv->n + 1, out, tmp.as<uint4>());
}

int acx_naive_columns(acx_naive* nv, int matrix, uint64_t wire_begin, uint64_t wire_count, acx_fr* out,
                      uint64_t* out_len) {
    ACX_RANGE();
    if (!nv || matrix < 0 || matrix > 2 || !out) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    acx_r1cs* r = nv->r;
    if (wire_begin + wire_count > r->m) return fail(ACX_ERR_INVALID_ARG, "wire range exceeds m");
    if (wire_count == 0) return ACX_OK;
    acx_ctx* c = r->ctx;
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    ACX_TRY(ensure_csc(r));
    const uint64_t N = 1ull << r->log_n, n = nv->n;
    DevBuf dense, res, tmp;
    ACX_TRY(dense.alloc(wire_count * N * 32));
    ACX_TRY(res.alloc(wire_count * n * 32));
    ACX_TRY(tmp.alloc(wire_count * n * 32));
    HIP_TRY(hipMemsetAsync(dense.p, 0, wire_count * N * 32, cur_stream(c)));
    const DevMatrix& T = r->T[matrix];
    if (T.nnz)
        hipLaunchKernelGGL(k_scatter_columns, dim3(grid_for(c, T.nnz)), dim3(kBlock), 0, cur_stream(c), (const u32*)T.ptr,
                           (const u32*)T.idx, (const u32*)T.colid, (const uint4*)T.val, wire_begin, wire_count, r->log_n, dense.as<uint4>());
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_matvec_q<F>), dim3(grid_for(c, wire_count * n)), dim3(kBlock), 0, cur_stream(c),
                                         (const uint4*)dense.as<uint4>(), N, (const uint4*)nv->Q, (u32)n, wire_count,
                                         res.as<uint4>(), n));
    HIP_TRY(hipGetLastError());
    ACX_TRY(download_elements(c, res.as<uint4>(), wire_count * n, out, tmp.as<uint4>()));
    if (out_len) {
        static const uint8_t zero32[32] = {0};
        for (uint64_t w = 0; w < wire_count; ++w) {
            uint64_t len = n;
            while (len > 0 && std::memcmp(out[w * n + len - 1].b, zero32, 32) == 0) --len;
            out_len[w] = len;
        }
    }
    return ACX_OK;
}

int acx_naive_h(acx_naive* nv, const acx_fr* witness, const acx_fr* delta, acx_fr* out_h, uint64_t* h_len, int* ok) {
    ACX_RANGE();
    if (!nv || !witness || !out_h || !h_len || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
    acx_r1cs* r = nv->r;
    acx_ctx* c = r->ctx;
    const HostField& hf = c->hf;
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    const uint64_t N = 1ull << r->log_n;
    const uint32_t n = nv->n, np1 = n + 1;
    DevBuf dots, lro, prod, quot;
    ACX_TRY(dots.alloc(3 * N * 32));
    ACX_TRY(lro.alloc((size_t)3 * np1 * 32));
    ACX_TRY(prod.alloc((size_t)(2 * np1) * 32));
    ACX_TRY(quot.alloc((size_t)(np1 + 1) * 32));
    HIP_TRY(hipMemsetAsync(dots.p, 0, 3 * N * 32, cur_stream(c)));
    HIP_TRY(hipMemsetAsync(lro.p, 0, (size_t)3 * np1 * 32, cur_stream(c)));
    HIP_TRY(hipMemsetAsync(quot.p, 0, (size_t)(np1 + 1) * 32, cur_stream(c)));
    uint64_t bad = 0;
    r->resident_valid = false;                                     // d_w doubles as this call's witness staging
    ACX_TRY(verify_common(r, witness, r->d_w, &bad, nullptr, nullptr, dots.as<uint4>(), N));
    H256 dl[3] = {hf.zero(), hf.zero(), hf.zero()};
    if (delta) for (int k = 0; k < 3; ++k) ACX_TRY(read_h256(&delta[k], hf, dl[k]));
    uint4* L = lro.as<uint4>();
    uint4* R = L + 2 * (u64)np1;
    uint4* O = R + 2 * (u64)np1;
    // L0, R0, O0 = interpolants of the dot products on the roots (n coefficients each, stride n+1)
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_matvec_q<F>), dim3(grid_for(c, 3ull * n)), dim3(kBlock), 0, cur_stream(c),
                                         (const uint4*)dots.as<uint4>(), N, (const uint4*)nv->Q, n, (u64)3, L, (u64)np1));
    // + delta_k * T   (src/QAP.hs:315-323)
    const FeArg one = dev_arg(hf, hf.one());
    for (int k = 0; k < 3; ++k)
        DISPATCH_FIELD(c, hipLaunchKernelGGL((k_poly_axpby<F>), dim3(grid_for(c, np1)), dim3(kBlock), 0, cur_stream(c),
                                             L + 2 * (u64)k * np1, (const uint4*)nv->tcoef, np1, one, dev_arg(hf, dl[k])));
    // P = L*R - O  (2n+1 coefficients), then quotRem by T
    DISPATCH_FIELD(c, {
        hipLaunchKernelGGL((k_poly_mul<F>), dim3((2 * np1 - 1 + kBlock - 1) / kBlock), dim3(kBlock), 0, cur_stream(c),
                           (const uint4*)L, np1, (const uint4*)R, np1, prod.as<uint4>());
        hipLaunchKernelGGL((k_poly_axpby<F>), dim3(grid_for(c, np1)), dim3(kBlock), 0, cur_stream(c), prod.as<uint4>(),
                           (const uint4*)O, np1, one, dev_arg(hf, hf.neg(hf.one())));
        hipLaunchKernelGGL((k_poly_divrem_monic<F>), dim3(1), dim3(1024), 0, cur_stream(c), prod.as<uint4>(), 2 * np1 - 1,
                           (const uint4*)nv->tcoef, n, quot.as<uint4>());
    });
    HIP_TRY(hipMemsetAsync(cur_err(c), 0, 4, cur_stream(c)));
    DISPATCH_FIELD(c, hipLaunchKernelGGL((k_any_nonzero<F>), dim3(grid_for(c, n)), dim3(kBlock), 0, cur_stream(c),
                                         (const uint4*)prod.as<uint4>(), n, cur_err(c)));
    HIP_TRY(hipGetLastError());
    uint32_t rem_nonzero = 0;
    HIP_TRY(hipMemcpyAsync(&rem_nonzero, cur_err(c), 4, hipMemcpyDeviceToHost, cur_stream(c)));
    ACX_TRY(download_elements(c, quot.as<uint4>(), np1, out_h, lro.as<uint4>()));
    *ok = rem_nonzero == 0;
    if ((bad == 0) != (*ok != 0)) return fail(ACX_ERR_HIP, "internal: division remainder disagrees with the residual check");
    uint64_t len = np1;
    static const uint8_t zero32[32] = {0};
    while (len > 0 && std::memcmp(out_h[len - 1].b, zero32, 32) == 0) --len;
    *h_len = len;
    return ACX_OK;
}

int acx_batch_create(acx_ctx* ctx, uint64_t count, acx_r1cs* const* systems, const void* const* d_witnesses,
                     uint64_t* d_results, uint64_t result_stride, acx_batch** out) {
    if (!ctx || !systems || !d_witnesses || !d_results || !out || count == 0 || count > 65535)
        return fail(ACX_ERR_INVALID_ARG, "bad batch arguments");
    CtxLock lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    acx_batch* b = new (std::nothrow) acx_batch();
    if (!b) return fail(ACX_ERR_OOM, "host allocation failed");
    b->ctx = ctx;
    std::vector<SellSystem> host(count);
    uint64_t row_offset = 0;
    for (uint64_t i = 0; i < count; ++i) {
        acx_r1cs* r = systems[i];
        if (!r || r->ctx != ctx || !d_witnesses[i]) { delete b; return fail(ACX_ERR_INVALID_ARG, "bad batch member"); }
        const ResidualOut o{(unsigned long long*)(d_results + i * result_stride), nullptr, nullptr, 0,
                            result_stride ? 0 : row_offset};
        host[i] = sell_system(r, (const uint4*)d_witnesses[i], o);
        b->systems.push_back(r);
        b->witnesses.push_back((const uint4*)d_witnesses[i]);
        b->outs.push_back(o);
        b->max_slices = std::max(b->max_slices, r->n_slices);
        row_offset += r->n;
    }
    if (hipMalloc((void**)&b->d_systems, count * sizeof(SellSystem)) != hipSuccess) { delete b; return fail(ACX_ERR_OOM, "device allocation failed"); }
    if (hipMemcpy(b->d_systems, host.data(), count * sizeof(SellSystem), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(b->d_systems); delete b; return fail(ACX_ERR_HIP, "descriptor upload failed");
    }
    *out = b;
    return ACX_OK;
}

void acx_batch_destroy(acx_batch* b) {
    if (!b) return;
    {
        CtxLock lock(b->ctx->mu);
        (void)hipSetDevice(b->ctx->device);
        (void)hipDeviceSynchronize();        // every lane: nothing may still be using this object
        if (b->d_systems) (void)hipFree(b->d_systems);
    }
    delete b;
}

int acx_batch_verify_dev(acx_batch* b) {
    ACX_RANGE();
    if (!b) return fail(ACX_ERR_INVALID_ARG, "null batch");
    acx_ctx* c = b->ctx;
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (b->max_slices) {
        const dim3 grid(sell_grid_x(b->max_slices), (unsigned)b->systems.size(), 1);
        int spec = sell_spec(b->systems[0]);
        for (const acx_r1cs* r : b->systems) spec = sell_spec_join(spec, sell_spec(r));
        launch_sell(c, spec, grid, b->d_systems, SellSystem{});
        HIP_TRY(hipGetLastError());
    }
    for (size_t i = 0; i < b->systems.size(); ++i)
        if (b->systems[i]->n_long) ACX_TRY(launch_long_rows(b->systems[i], b->witnesses[i], b->outs[i]));
    return ACX_OK;
}

int acx_ntt_dev(acx_ctx* c, uint32_t log_n, uint64_t batch, int inverse, const acx_fr* shift, void* d_data) {
    ACX_RANGE();
    if (!c || !d_data) return fail(ACX_ERR_INVALID_ARG, "null argument");
    CtxLock lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    H256 sh;
    if (shift) {
        ACX_TRY(read_h256(shift, c->hf, sh));
        if (sh.is_zero()) return fail(ACX_ERR_INVALID_ARG, "coset shift must be nonzero");
    }
    return ntt_dev_locked(c, (uint4*)d_data, log_n, batch, inverse, shift ? &sh : nullptr);
}

}  // extern "C"

#ifdef ACX_K2_TRACE
// development build only (tools/k2_trace.py): read and clear the residual kernel's phase accumulators
extern "C" int acx_debug_k2_trace(unsigned long long out[16]) {
    static std::vector<unsigned long long> host(2ull * acx::kK2TraceWaves * 6);
    if (hipMemcpyFromSymbol(host.data(), HIP_SYMBOL(acx::g_k2_trace), host.size() * 8) != hipSuccess) return ACX_ERR_HIP;
    for (int r = 0; r < 2; ++r) {
        for (int k = 0; k < 8; ++k) out[8 * r + k] = 0;
        for (uint64_t i = 0; i < acx::kK2TraceWaves; ++i)
            for (int k = 0; k < 6; ++k) out[8 * r + k] += host[((uint64_t)r * acx::kK2TraceWaves + i) * 6 + k];
    }
    std::fill(host.begin(), host.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(acx::g_k2_trace), host.data(), host.size() * 8) != hipSuccess) return ACX_ERR_HIP;
    return ACX_OK;
}
#endif

#include "mgpu.inc.h"
