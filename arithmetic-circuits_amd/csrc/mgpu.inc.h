// mgpu.inc.h -- acx_mgpu_*: ONE process, N GPUs, behind the C ABI (included at the end of engine.hip).
//
// The reference's callers are one thread making one pure call -- `verifyAssignment qap assignment`
// (/root/reference/src/QAP.hs:276-282), `all (verifyAssignment qap . generateAssignment program) inputs`
// (test/Test/Circuit/Arithmetic.hs:200-209), `verificationWitness` (src/QAP.hs:292-327) -- so the sharding over the GPUs
// of a node and the collectives between them live HERE, under the header, not in the host program:
//
//   rows          TWO ownerships per system (288 GB of HBM per GPU: memory is not what is scarce).  For verifyAssignment:
//                 contiguous slabs balanced by nnz -- the rows in flight on a GPU gather from one narrow
//                 window of the witness, which is worth 1.5-2x on the residual kernel (profiles/r02_dist_budget.txt).  For
//                 h(x): block-cyclic -- with N = 2^log_n = R * C, shard g owns the rows k = k1 + k2 R with k1 in block g of R/W,
//                 held in ASCENDING order (runs of R/W consecutive rows: the gathers of the rows in flight stay inside a
//                 window 8x narrower than in ROWS order [kl][k2], profiles/r02_dist_budget.txt), so its residual kernel
//                 writes <A_i,w>, <B_i,w>, <C_i,w> as the transposed ROWS block [k2][kl], which the first inverse step reads
//                 through its strides (ntt_dist_step_locked, rows_transposed).  Rows >= n are empty rows.
//                 ACX_MGPU_VERIFY_ONLY at load time skips the second copy.
//   witness       replicated: one host-to-device copy per GPU (each over its own PCIe link, one host thread each)
//   verdict       ONE ncclAllReduce (sum of the violated-row counts); a second one (min) only for first_bad of a failing check
//   transforms    four-step, one launch per local step (ntt_dist_step_locked) and ONE ncclAllToAll between the two steps, issued
//                 on a second stream per GPU so that vector k's exchange runs under vector k+1's local step
//   h(x)          3 inverse + 2 coset + pointwise + 1 inverse coset transform, minus O / z in coefficient form (six
//                 all-to-alls: qap_h_dev_locked's pipeline, distributed), h gathered into natural order by strided copies
//
// Transport.  RCCL (ncclCommInitAll over the device list; bound with dlopen at acx_mgpu_create, so single-GPU users of
// libacx never map the RCCL library) whenever the device ids are distinct.  A device list with REPEATED ids -- several
// shards on one GPU: how the W = 2 / 4 / 8 code paths run on a one-GPU box -- cannot form an RCCL communicator; the exchange
// is then W x W peer copies (hipMemcpyPeerAsync, pulled by the receiving shard's exchange stream) and the verdict is summed
// on the host.  ACX_MGPU_TRANSPORT=peer selects the copies on distinct devices too (xGMI DMA engines instead of RCCL's
// kernels: no CUs taken from the local steps).  Same events, same buffers, same results either way.
//
// One host thread issues everything (the documented single-process RCCL pattern: ncclGroupStart / per-device calls /
// ncclGroupEnd); every call is asynchronous on the per-GPU streams, so the host runs ahead of the devices.  Program order on
// that one thread is also what makes the cross-stream events safe (hipStreamWaitEvent on an event not yet recorded is a
// no-op; here every wait is issued after its record).
#include <dlfcn.h>
#include <link.h>
#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <thread>

namespace {

// ---- RCCL, bound at run time ------------------------------------------------------------------------
struct RcclApi {
    void* so = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*AllToAll)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

// An RCCL already mapped into the process (a Python host with torch has torch's own copy) is the one to use: two RCCL
// copies would each bring their own kernels and state for the same devices.
static int find_loaded_rccl(struct dl_phdr_info* info, size_t, void* out) {
    if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) {
        *static_cast<std::string*>(out) = info->dlpi_name;
        return 1;
    }
    return 0;
}

static const RcclApi* rccl_api(std::string& why) {
    static std::mutex mu;
    static RcclApi api;
    static bool tried = false;
    static std::string err;
    std::lock_guard<std::mutex> g(mu);
    if (!tried) {
        tried = true;
        std::string loaded;
        dl_iterate_phdr(find_loaded_rccl, &loaded);
        const char* names[] = {loaded.empty() ? nullptr : loaded.c_str(), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm) continue;
            api.so = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.so) break;
        }
        if (!api.so) {
            const char* e = dlerror();                                  // ONE call: dlerror() clears the message it returns
            err = std::string("RCCL not found (dlopen librccl.so.1): ") + (e ? e : "");
        } else {
            auto sym = [&](const char* name) { void* p = dlsym(api.so, name); if (!p && err.empty()) err = std::string("RCCL symbol missing: ") + name; return p; };
            api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
            api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
            api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
            api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
            api.AllToAll = reinterpret_cast<decltype(api.AllToAll)>(sym("ncclAllToAll"));
            api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
            api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
            api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        }
    }
    if (!err.empty()) { why = err; return nullptr; }
    return &api;
}

#define NCCL_TRY(mg, expr)                                                                                   \
    do {                                                                                                     \
        ncclResult_t r_ = (expr);                                                                            \
        if (r_ != ncclSuccess) return fail(ACX_ERR_HIP, std::string(#expr) + ": " + (mg)->api->GetErrorString(r_)); \
    } while (0)

// ---- one persistent issuing thread per shard ------------------------------------------------------------
// A call on the handle is W independent streams of API calls (launches, event records and waits, copies: ~50 per shard per
// h(x)).  Issued from one thread they are serial -- W x 50 calls of 2-5 us each against a few milliseconds of device time --
// so every shard has its own host thread for the life of the handle (no thread creation per call either: that alone was
// 20-50 us per shard).  run(fn) hands fn(shard) to every worker and returns when all are done; the first failure (and the
// failing thread's message) is carried back.  With RCCL each thread drives its own communicator (the documented
// multi-threaded single-process pattern) and nothing crosses threads on the host.  With the peer-copy transport a shard waits
// on events its PEERS record, and a wait on an event not yet recorded is a no-op: barrier() orders those host-side
// (a worker that has failed releases the barrier for everybody: no deadlock on the error path).
struct MgPool {
    uint32_t W = 0;
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    const std::function<int(uint32_t)>* job = nullptr;
    uint64_t gen = 0;
    uint32_t pending = 0;
    bool stop = false;
    std::vector<int> rc;
    std::vector<std::string> msg;
    std::mutex bmu;
    std::condition_variable bcv;
    uint32_t bcount = 0;
    uint64_t bgen = 0;
    bool aborted = false;

    void start(uint32_t w, const std::vector<int>& devices) {
        W = w;
        rc.assign(W, ACX_OK);
        msg.assign(W, std::string());
        for (uint32_t s = 0; s < W; ++s) th.emplace_back([this, s, dev = devices[s]] { loop(s, dev); });
    }
    void loop(uint32_t s, int device) {
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
            const std::function<int(uint32_t)>* fn = nullptr;
            {
                std::unique_lock<std::mutex> l(mu);
                cv_go.wait(l, [&] { return stop || gen != seen; });
                if (stop) return;
                seen = gen;
                fn = job;
            }
            int r = ACX_OK;
            try {
                (void)hipSetDevice(device);
                r = (*fn)(s);
            } catch (const std::bad_alloc&) {
                r = fail(ACX_ERR_OOM, "host allocation failed");
            } catch (...) {
                r = fail(ACX_ERR_INVALID_ARG, "unexpected exception");
            }
            if (r != ACX_OK) {
                std::lock_guard<std::mutex> b(bmu);
                aborted = true;
                bcv.notify_all();
            }
            std::lock_guard<std::mutex> l(mu);
            rc[s] = r;
            if (r != ACX_OK) msg[s] = g_last_error;
            if (--pending == 0) cv_done.notify_all();
        }
    }
    int run(const std::function<int(uint32_t)>& fn) {
        {
            std::lock_guard<std::mutex> b(bmu);
            aborted = false;
            bcount = 0;
        }
        std::unique_lock<std::mutex> l(mu);
        job = &fn;
        pending = W;
        ++gen;
        cv_go.notify_all();
        cv_done.wait(l, [&] { return pending == 0; });
        for (uint32_t s = 0; s < W; ++s)
            if (rc[s] != ACX_OK) return fail(rc[s], msg[s]);
        return ACX_OK;
    }
    // every worker of the current job; false: another worker failed, give up
    bool barrier() {
        std::unique_lock<std::mutex> l(bmu);
        if (aborted) return false;
        const uint64_t my = bgen;
        if (++bcount == W) {
            bcount = 0;
            ++bgen;
            bcv.notify_all();
            return true;
        }
        bcv.wait(l, [&] { return aborted || bgen != my; });
        return !aborted;
    }
    void shutdown() {
        {
            std::lock_guard<std::mutex> l(mu);
            stop = true;
        }
        cv_go.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
        th.clear();
    }
};

// The receiving side of the peer-copy exchange for the sources that live on the receiver's OWN device (a device list with a
// repeated ordinal: several shards on one GPU): block t of recv = block `me` of shard t's send buffer, every t in one launch
// instead of one device-to-device copy per source (W x W copies per exchange made the one-GPU configuration host bound).
struct MgPull {
    const uint4* src[64];                           // send + me * chunk of every same-device source; null: not on this device
};
__global__ __launch_bounds__(kBlock) void k_pull_chunks(MgPull P, uint4* __restrict__ recv, u32 W, u64 chunk_quads) {
    const u32 t = blockIdx.y;
    if (t >= W || P.src[t] == nullptr) return;
    const uint4* from = P.src[t];
    uint4* to = recv + (u64)t * chunk_quads;
    for (u64 i = (u64)blockIdx.x * kBlock + threadIdx.x; i < chunk_quads; i += (u64)gridDim.x * kBlock) to[i] = gload(from + i);
}

constexpr int kMgSlots = 3;          // transforms in flight (the three vectors of h(x))
constexpr uint32_t kMgRing = 16;     // result slots of acx_mgpu_r1cs_verify_enqueue
constexpr uint64_t kMgColBlock = 64; // wires per block of the block-cyclic WIRE ownership of acx_mgpu_qap_columns

struct MgSlot {                      // one exchange buffer pair of one shard
    uint4 *send = nullptr, *recv = nullptr;
    hipEvent_t sent = nullptr;       // recorded on the compute stream after step 0 (send is complete)
    hipEvent_t got = nullptr;        // recorded on the exchange stream after the exchange (recv is complete; peers' pulls of
                                     // this shard's send happen on THEIR exchange streams, see mg_exchange)
    hipEvent_t used = nullptr;       // recorded on the compute stream after step 1 (recv may be overwritten)
    bool got_valid = false, used_valid = false;
};

struct MgShard {
    acx_ctx* ctx = nullptr;
    int device = 0;
    hipStream_t xstream = nullptr;                  // exchanges
    ncclComm_t comm = nullptr;
    unsigned long long* d_res = nullptr;            // CallSlot {n_bad, first_bad, non-canonical flag} + 2 reduction words
    MgSlot slot[kMgSlots];
    uint64_t slot_elems = 0;
    uint4* io = nullptr;                            // staging of the natural-order host transfers (acx_mgpu_ntt, h fetch)
    uint64_t io_elems = 0;
    hipEvent_t w_ready = nullptr;                   // shard 0: the converted witness is complete (peers pull it)
    hipEvent_t w_read = nullptr;                    // other shards: their copy out of shard 0's buffer is done
    bool w_read_valid = false;
};

}  // namespace

struct acx_mgpu {
    int field = 0;
    uint32_t W = 0;
    bool rccl = false;
    const RcclApi* api = nullptr;
    std::vector<MgShard> sh;
    uint32_t min_log_n = 14;                        // smaller systems stay on shard 0 (acx_mgpu_set_shard_threshold)
    std::mutex mu;                                  // one acx_mgpu_* call at a time: collectives are ordered
    std::unique_ptr<MgPool> pool;                   // W > 1: one issuing thread per shard
    int witness_mode = 0;                           // 0 broadcast (one H2D + device-side replication), 1 W host copies, 2 the same from registered memory
    // wall clock of the last verify / h(x) call on this handle: entry -> everything enqueued (the HOST's share: API calls
    // of the issuing threads) and entry -> results on the host.  acx_mgpu_debug_times (tools/mgpu_host.py).
    double last_issue_s = 0, last_total_s = 0;
};

struct acx_mgpu_r1cs {
    acx_mgpu* mg = nullptr;
    uint64_t n = 0, m = 0;
    uint32_t log_n = 0, log_r = 0;
    bool sharded = false;
    acx_r1cs* whole = nullptr;                      // !sharded: the whole system on shard 0
    struct Part {
        acx_r1cs* slab = nullptr;                   // rows [row0, row0 + slab->n): what verifyAssignment runs on
        uint64_t row0 = 0;
        acx_r1cs* cyc = nullptr;                    // this shard's N/W block-cyclic rows in ascending order: what h(x) runs on (null: verify only)
        acx_r1cs* full = nullptr;                   // the WHOLE system (shard 0 only, on demand): acx_mgpu_qap_h of a size the four-step form does not cover
        acx_r1cs* cols = nullptr;                   // the column view of THIS shard's wires (block-cyclic, kMgColBlock wires per block): acx_mgpu_qap_columns
        uint4* d_w = nullptr;                       // the replicated witness, m dev elements
        uint4* vec = nullptr;                       // h(x) pipeline: dots 3L | coef 3L | pw L | h L (allocated on first use)
        uint4* hscale = nullptr;                    // {1/z, -1/z} for the GLOBAL N as dev elements: ride on the stored dots of h(x) (qap_h_dev_locked)
        unsigned long long* ring = nullptr;         // four sections of kMgRing result slots {n_bad, first_bad}: the asynchronous form's slots,
                                                    // their reduction, acx_mgpu_r1cs_verify_many's own slots, their reduction
    };
    std::vector<Part> part;
    bool has_cyclic = false;
    bool verify_only = false;                       // loaded with ACX_MGPU_VERIFY_ONLY: h(x) is refused, not computed some other way
    bool witness_resident = false;
    bool h_valid = false;                           // part[].vec holds h of the resident witness (acx_mgpu_qap_h_fetch)
    H256 h_top{{0, 0, 0, 0}};                       // coefficient N of the zero-knowledge quotient (d1 d2), Montgomery
};

namespace {

struct MgClock {                                    // issue / total wall clock of one call, written to the handle on exit
    acx_mgpu* mg;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double issue = -1;
    explicit MgClock(acx_mgpu* m) : mg(m) {}
    void issued() { issue = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
    ~MgClock() {
        mg->last_total_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mg->last_issue_s = issue < 0 ? mg->last_total_s : issue;
    }
};

struct DevGuard {                                   // the calling thread's device is restored on exit
    int prev = 0;
    DevGuard() { (void)hipGetDevice(&prev); }
    ~DevGuard() { (void)hipSetDevice(prev); }
};

inline uint32_t mg_log2(uint32_t w) { uint32_t k = 0; while ((1u << k) < w) ++k; return k; }

// can a 2^log_n-point transform be spread over W shards?  (5 <= both digits <= 12, W divides both, an odd digit needs
// two local columns)
inline bool mg_can_distribute(uint32_t W, uint32_t log_n) {
    if (log_n < 10 || log_n > 24) return false;
    const uint32_t lr = log_n / 2, lc = log_n - lr, lw = mg_log2(W);
    return lr >= lw + 1 && lc >= lw + 1;
}

int mg_ensure_slots(acx_mgpu* mg, uint64_t L) {
    for (auto& s : mg->sh) {
        HIP_TRY(hipSetDevice(s.device));
        if (s.slot_elems >= L) continue;
        HIP_TRY(hipDeviceSynchronize());
        for (auto& sl : s.slot) {
            if (sl.send) (void)hipFree(sl.send);
            if (sl.recv) (void)hipFree(sl.recv);
            sl.send = sl.recv = nullptr;
            sl.got_valid = sl.used_valid = false;
        }
        s.slot_elems = 0;
        for (auto& sl : s.slot) {
            HIP_TRY(hipMalloc((void**)&sl.send, L * 32));
            HIP_TRY(hipMalloc((void**)&sl.recv, L * 32));
        }
        s.slot_elems = L;
    }
    return ACX_OK;
}

int mg_ensure_io(acx_mgpu* mg, uint64_t L) {
    for (auto& s : mg->sh) {
        HIP_TRY(hipSetDevice(s.device));
        if (s.io_elems >= L) continue;
        HIP_TRY(hipDeviceSynchronize());
        if (s.io) (void)hipFree(s.io);
        s.io = nullptr; s.io_elems = 0;
        HIP_TRY(hipMalloc((void**)&s.io, L * 32));
        s.io_elems = L;
    }
    return ACX_OK;
}

// run fn(shard) for every shard, each on its shard's own persistent host thread (MgPool); the first failure and its message
// are carried back to the calling thread.  One shard: on the calling thread.
template <class Fn>
int mg_per_shard_threads(acx_mgpu* mg, Fn&& fn) {
    if (mg->W == 1 || !mg->pool) {
        for (uint32_t s = 0; s < mg->W; ++s) {
            int rc;
            try {
                rc = fn(s);
            } catch (const std::bad_alloc&) {
                rc = fail(ACX_ERR_OOM, "host allocation failed");
            } catch (...) {
                rc = fail(ACX_ERR_INVALID_ARG, "unexpected exception");
            }
            if (rc != ACX_OK) return rc;
        }
        return ACX_OK;
    }
    const std::function<int(uint32_t)> f = std::forward<Fn>(fn);
    return mg->pool->run(f);
}
inline bool mg_barrier(acx_mgpu* mg) { return mg->W == 1 || !mg->pool || mg->pool->barrier(); }
#define MG_BARRIER(mg)                                                                                       \
    do {                                                                                                     \
        if (!mg_barrier(mg)) return fail(ACX_ERR_HIP, "another shard's issuing thread failed");               \
    } while (0)

// ---- one distributed transform = begin (local step 0 + the START of the exchange) and finish (wait + local step 1) ----
// Every method is the part of ONE shard, called by that shard's issuing thread (mg_per_shard_threads): the W threads run the
// same sequence, so their barriers (peer-copy transport only) pair up.
struct MgNtt {
    acx_mgpu* mg;
    uint32_t log_n, log_r;
    uint64_t L, chunk;                              // elements per shard; per (shard, peer) block
    MgNtt(acx_mgpu* m, uint32_t ln, uint32_t lr) : mg(m), log_n(ln), log_r(lr) {
        L = (1ull << ln) / m->W;
        chunk = L / m->W;
    }

    int exchange(uint32_t s, int k) {
        const uint32_t W = mg->W;
        MgShard& S = mg->sh[s];
        MgSlot& sl = S.slot[k];
        if (mg->rccl) {                                             // this shard's rank of THE all-to-all, on its own communicator
            HIP_TRY(hipStreamWaitEvent(S.xstream, sl.sent, 0));
            if (sl.used_valid) HIP_TRY(hipStreamWaitEvent(S.xstream, sl.used, 0));            // previous reader of recv
            NCCL_TRY(mg, mg->api->AllToAll(sl.send, sl.recv, chunk * 4, ncclUint64, S.comm, S.xstream));
            HIP_TRY(hipEventRecord(sl.got, S.xstream));
            sl.got_valid = true;
            return ACX_OK;
        }
        // peer copies: this shard PULLS its block of every shard's send buffer.  The waits below are on events the peers'
        // threads record: all of them must have been recorded first (a wait on an unrecorded event is a no-op).
        MG_BARRIER(mg);
        if (sl.used_valid) HIP_TRY(hipStreamWaitEvent(S.xstream, sl.used, 0));
        MgPull pull{};
        bool local = false;
        for (uint32_t t = 0; t < W; ++t) {
            MgShard& src = mg->sh[t];
            HIP_TRY(hipStreamWaitEvent(S.xstream, src.slot[k].sent, 0));
            const uint4* from = src.slot[k].send + 2 * (uint64_t)s * chunk;
            if (src.device == S.device) { pull.src[t] = from; local = true; }
            else HIP_TRY(hipMemcpyPeerAsync(sl.recv + 2 * (uint64_t)t * chunk, S.device, from, src.device, chunk * 32, S.xstream));
        }
        if (local) {
            const unsigned gx = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>((2 * chunk + kBlock - 1) / kBlock, 4ull * S.ctx->n_cu / W + 1));
            hipLaunchKernelGGL(k_pull_chunks, dim3(gx, W), dim3(kBlock), 0, S.xstream, pull, sl.recv, W, (u64)(2 * chunk));
            HIP_TRY(hipGetLastError());
        }
        HIP_TRY(hipEventRecord(sl.got, S.xstream));
        sl.got_valid = true;
        MG_BARRIER(mg);                                             // every `got` of this exchange is recorded: the next begin on slot k may wait on them
        return ACX_OK;
    }

    // in: L dev elements of shard s (COLS for a forward, ROWS for an inverse transform)
    // rows_transposed: the input of an inverse transform is in ascending row order [k2][kl] (the residual kernel's dots)
    // mul: the transform of the pointwise product in[i] * mul[i] (same layout)
    int begin(uint32_t s, int k, const uint4* in, int inverse, const H256* shift, bool rows_transposed = false, const uint4* mul = nullptr) {
        const uint32_t W = mg->W;
        MgShard& S = mg->sh[s];
        {
            CtxLock lock(S.ctx->mu);
            if (!mg->rccl)                                          // peers still pulling the previous contents of send
                for (uint32_t t = 0; t < W; ++t)
                    if (mg->sh[t].slot[k].got_valid) HIP_TRY(hipStreamWaitEvent(S.ctx->stream, mg->sh[t].slot[k].got, 0));
            ACX_TRY(ntt_dist_step_locked(S.ctx, log_n, log_r, W, s, inverse, 0, shift, in, S.slot[k].send, rows_transposed, mul));
            HIP_TRY(hipEventRecord(S.slot[k].sent, S.ctx->stream));
        }
        return exchange(s, k);
    }

    // add: out[k] = X[k] + add[k] (same layout as out)
    int finish(uint32_t s, int k, uint4* out, int inverse, const H256* shift, const uint4* add = nullptr) {
        MgShard& S = mg->sh[s];
        CtxLock lock(S.ctx->mu);
        HIP_TRY(hipStreamWaitEvent(S.ctx->stream, S.slot[k].got, 0));
        ACX_TRY(ntt_dist_step_locked(S.ctx, log_n, log_r, mg->W, s, inverse, 1, shift, S.slot[k].recv, out, false, nullptr, add));
        HIP_TRY(hipEventRecord(S.slot[k].used, S.ctx->stream));
        S.slot[k].used_valid = true;
        return ACX_OK;
    }
};

// Replicate the witness on every shard (dev format; the canonicity flag lands in shard 0's CallSlot, and in every shard's with
// the host-copy modes).  mg->witness_mode (ACX_MGPU_WITNESS = broadcast | copies | pinned):
//   broadcast  ONE host-to-device copy and ONE conversion, on shard 0; the other shards receive the converted elements over
//              the device fabric -- ncclBroadcast on every shard's stream (xGMI), or one device copy each pulled by the shard
//              itself with the peer-copy transport.  m * 32 bytes cross PCIe once instead of W times.
//   copies     one pageable host-to-device copy per shard, each from its own host thread over its own PCIe link
//   pinned     the same from registered memory: the caller's buffer is page-locked for the duration of the call
int mg_upload_witness(acx_mgpu_r1cs* mr, const acx_fr* witness) {
    acx_mgpu* mg = mr->mg;
    mr->witness_resident = false;
    mr->h_valid = false;
    const uint32_t W = mg->W;
    const int mode = W == 1 ? 1 : mg->witness_mode;
    static const CallSlot init{0ull, ~0ull, 0u, {0u, 0u, 0u}};
    bool registered = false;
    if (mode == 2) registered = hipHostRegister(const_cast<acx_fr*>(witness), mr->m * 32, hipHostRegisterDefault) == hipSuccess;
    if (mode == 2 && !registered) (void)hipGetLastError();          // e.g. already registered by the caller: plain copies then
    const int rc = mg_per_shard_threads(mg, [&](uint32_t s) -> int {
        MgShard& S = mg->sh[s];
        HIP_TRY(hipSetDevice(S.device));
        CtxLock lock(S.ctx->mu);
        HIP_TRY(hipMemcpyAsync(S.d_res, &init, sizeof(init), hipMemcpyHostToDevice, S.ctx->stream));
        uint4* d_w = mr->part[s].d_w;
        if (mode != 0) {
            HIP_TRY(hipMemcpyAsync(d_w, witness, mr->m * 32, hipMemcpyHostToDevice, S.ctx->stream));
            return launch_convert(S.ctx, true, d_w, d_w, mr->m, (uint32_t*)(S.d_res + 2));
        }
        if (s == 0) {
            if (!mg->rccl)                                          // peers still reading the previous witness out of shard 0's buffer
                for (uint32_t t = 1; t < W; ++t)
                    if (mg->sh[t].w_read_valid) HIP_TRY(hipStreamWaitEvent(S.ctx->stream, mg->sh[t].w_read, 0));
            HIP_TRY(hipMemcpyAsync(d_w, witness, mr->m * 32, hipMemcpyHostToDevice, S.ctx->stream));
            ACX_TRY(launch_convert(S.ctx, true, d_w, d_w, mr->m, (uint32_t*)(S.d_res + 2)));
            if (!mg->rccl) HIP_TRY(hipEventRecord(S.w_ready, S.ctx->stream));
        }
        if (mg->rccl)
            // in place on every rank: the root sends its own buffer, the others' send pointer is unused (and stays a pointer of
            // THEIR device, whatever pointer checks the collective library applies)
            NCCL_TRY(mg, mg->api->Broadcast(d_w, d_w, mr->m * 4, ncclUint64, 0, S.comm, S.ctx->stream));
        else {
            MG_BARRIER(mg);                                         // shard 0's w_ready is recorded
            if (s != 0) {
                MgShard& S0 = mg->sh[0];
                HIP_TRY(hipStreamWaitEvent(S.ctx->stream, S0.w_ready, 0));
                if (S0.device == S.device) HIP_TRY(hipMemcpyAsync(d_w, mr->part[0].d_w, mr->m * 32, hipMemcpyDeviceToDevice, S.ctx->stream));
                else HIP_TRY(hipMemcpyPeerAsync(d_w, S.device, mr->part[0].d_w, S0.device, mr->m * 32, S.ctx->stream));
                HIP_TRY(hipEventRecord(S.w_read, S.ctx->stream));
                S.w_read_valid = true;
            }
        }
        return ACX_OK;
    });
    if (registered) {
        for (auto& S : mg->sh) { (void)hipSetDevice(S.device); (void)hipStreamSynchronize(S.ctx->stream); }
        (void)hipHostUnregister(const_cast<acx_fr*>(witness));
    }
    ACX_TRY(rc);
    mr->witness_resident = true;
    return ACX_OK;
}

// residual launch on every shard (+ dots when the h(x) pipeline follows) and the verdict.
// Two halves, so that h(x) can issue its whole pipeline between them and the host waits once, at the end.
// mg_residual_enqueue_shard: ONE shard's residual launch (+ dots when the h(x) pipeline follows) and, with RCCL, its rank of THE
// verdict collective behind it -- everything asynchronous, on the shard's issuing thread.  mg_residual_fetch: the verdict (one wait).
int mg_residual_enqueue_shard(acx_mgpu_r1cs* mr, uint32_t s, bool with_dots, bool scaled_dots) {
    acx_mgpu* mg = mr->mg;
    const uint32_t W = mg->W;
    const uint64_t L = (1ull << mr->log_n) / W, rw = (1ull << mr->log_r) / W;
    MgShard& S = mg->sh[s];
    HIP_TRY(hipSetDevice(S.device));
    CtxLock lock(S.ctx->mu);
    static const unsigned long long init[2] = {0ull, ~0ull};
    HIP_TRY(hipMemcpyAsync(S.d_res, init, 16, hipMemcpyHostToDevice, S.ctx->stream));      // the canonicity flag stays
    const auto& P = mr->part[s];
    if (with_dots)          // the block-cyclic copy: dots in ascending row order (= ROWS transposed), first_bad through the run map
        ACX_TRY(launch_residual(P.cyc, P.d_w, (uint64_t)s * rw, S.d_res, nullptr, P.vec, L, mr->log_r - mg_log2(W), mr->log_r,
                                scaled_dots ? (const uint4*)P.hscale : nullptr));
    else
        ACX_TRY(launch_residual(P.slab, P.d_w, P.row0, S.d_res, nullptr, nullptr, 0));
    if (mg->rccl)           // THE verdict collective: sum of the violated-row counts, into word 4 of every shard's slot
        NCCL_TRY(mg, mg->api->AllReduce(S.d_res, S.d_res + 4, 1, ncclUint64, ncclSum, S.comm, S.ctx->stream));
    return ACX_OK;
}

int mg_residual_enqueue(acx_mgpu_r1cs* mr, bool with_dots, bool scaled_dots = false) {
    return mg_per_shard_threads(mr->mg, [&](uint32_t s) -> int { return mg_residual_enqueue_shard(mr, s, with_dots, scaled_dots); });
}

int mg_residual_fetch(acx_mgpu_r1cs* mr, bool want_first, uint64_t* n_bad, uint64_t* first_bad, bool* noncanonical) {
    acx_mgpu* mg = mr->mg;
    const uint32_t W = mg->W;
    CallSlot slot0;
    unsigned long long total = 0, first = ~0ull;
    if (mg->rccl) {
        MgShard& S0 = mg->sh[0];
        HIP_TRY(hipSetDevice(S0.device));
        HIP_TRY(hipMemcpyAsync(&slot0, S0.d_res, sizeof(slot0), hipMemcpyDeviceToHost, S0.ctx->stream));
        HIP_TRY(hipMemcpyAsync(&total, S0.d_res + 4, 8, hipMemcpyDeviceToHost, S0.ctx->stream));
        HIP_TRY(hipStreamSynchronize(S0.ctx->stream));
        if (total != 0 && want_first) {                             // on request, and only for a failing check
            NCCL_TRY(mg, mg->api->GroupStart());
            for (auto& S : mg->sh) {
                const ncclResult_t r = mg->api->AllReduce(S.d_res + 1, S.d_res + 5, 1, ncclUint64, ncclMin, S.comm, S.ctx->stream);
                if (r != ncclSuccess) { (void)mg->api->GroupEnd(); return fail(ACX_ERR_HIP, std::string("ncclAllReduce: ") + mg->api->GetErrorString(r)); }
            }
            NCCL_TRY(mg, mg->api->GroupEnd());
            HIP_TRY(hipMemcpyAsync(&first, S0.d_res + 5, 8, hipMemcpyDeviceToHost, S0.ctx->stream));
            HIP_TRY(hipStreamSynchronize(S0.ctx->stream));
        }
    } else {
        std::vector<CallSlot> slots(W);
        for (uint32_t s = 0; s < W; ++s) {
            MgShard& S = mg->sh[s];
            HIP_TRY(hipSetDevice(S.device));
            HIP_TRY(hipMemcpyAsync(&slots[s], S.d_res, sizeof(CallSlot), hipMemcpyDeviceToHost, S.ctx->stream));
        }
        for (uint32_t s = 0; s < W; ++s) {
            HIP_TRY(hipSetDevice(mg->sh[s].device));
            HIP_TRY(hipStreamSynchronize(mg->sh[s].ctx->stream));
            total += slots[s].n_bad;
            first = std::min<unsigned long long>(first, slots[s].first_bad);
        }
        slot0 = slots[0];
    }
    *noncanonical = slot0.noncanonical != 0;
    *n_bad = total;
    *first_bad = (total != 0 && want_first) ? first : ~0ull;
    return ACX_OK;
}

int mg_residual(acx_mgpu_r1cs* mr, bool with_dots, bool want_first, uint64_t* n_bad, uint64_t* first_bad, bool* noncanonical, MgClock* clock = nullptr) {
    ACX_TRY(mg_residual_enqueue(mr, with_dots));
    if (clock) clock->issued();
    return mg_residual_fetch(mr, want_first, n_bad, first_bad, noncanonical);
}

// [rows][cols] -> [cols][rows] of 32-byte elements through a 32 x 32 LDS tile, with the canonical <-> dev conversion of the
// host edge fused (MODE 0 none, 1 canonical -> dev with the canonicity check, 2 dev -> canonical).
template <class F, int MODE>
__global__ __launch_bounds__(kBlock) void k_transpose(const uint4* __restrict__ in, uint4* __restrict__ out, u32 rows, u32 cols,
                                                     u32* __restrict__ err) {
    __shared__ uint4 tile[32][2 * 32 + 1];
    const u32 tiles_c = (cols + 31) / 32;
    const u32 tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
    for (u32 i = threadIdx.x; i < 1024; i += kBlock) {
        const u32 r = tr * 32 + i / 32, c = tc * 32 + i % 32;
        if (r < rows && c < cols) {
            Fe x = fe_load(in + 2 * ((u64)r * cols + c));
            if (MODE == 1) {
                if (err != nullptr && !fe_lt_p<F>(x)) atomicOr(err, 1u);
                x = fe_to_mont<F>(x);
            } else if (MODE == 2) {
                x = fe_from_mont<F>(x);
            }
            u32 w[8];
            fe_pack(x, w);
            tile[i / 32][2 * (i % 32)] = make_uint4(w[0], w[1], w[2], w[3]);
            tile[i / 32][2 * (i % 32) + 1] = make_uint4(w[4], w[5], w[6], w[7]);
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < 1024; i += kBlock) {
        const u32 c = tc * 32 + i / 32, r = tr * 32 + i % 32;          // output row = input column
        if (r < rows && c < cols) {
            out[2 * ((u64)c * rows + r)] = tile[i % 32][2 * (i / 32)];
            out[2 * ((u64)c * rows + r) + 1] = tile[i % 32][2 * (i / 32) + 1];
        }
    }
}

int mg_transpose(acx_ctx* c, int mode, const uint4* in, uint4* out, uint64_t rows, uint64_t cols, uint32_t* d_err) {
    const unsigned grid = (unsigned)(((rows + 31) / 32) * ((cols + 31) / 32));
    DISPATCH_FIELD(c, {
        if (mode == 1) hipLaunchKernelGGL((k_transpose<F, 1>), dim3(grid), dim3(kBlock), 0, c->stream, in, out, (u32)rows, (u32)cols, d_err);
        else if (mode == 2) hipLaunchKernelGGL((k_transpose<F, 2>), dim3(grid), dim3(kBlock), 0, c->stream, in, out, (u32)rows, (u32)cols, d_err);
        else hipLaunchKernelGGL((k_transpose<F, 0>), dim3(grid), dim3(kBlock), 0, c->stream, in, out, (u32)rows, (u32)cols, d_err);
    });
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

// Natural-order host vector <-> the shards' blocks.  COLS [i2l][i1] holds x[i1*C + g*C/W + i2l]; ROWS [kl][k2] holds
// X[(g*R/W + kl) + k2*R] (include/acx.h).  Either is "outer index o (count P), runs of q elements at g*q + o*stride":
// COLS: P = R, q = C/W, stride = C; ROWS: P = C, q = R/W, stride = R -- stored transposed, [q][P].
// download: transpose + dev -> canonical on the device, then ONE strided device-to-host copy per shard.
int mg_fetch_natural(acx_mgpu* mg, uint4* const* d_blocks, uint64_t P, uint64_t q, uint64_t stride, acx_fr* host) {
    for (uint32_t s = 0; s < mg->W; ++s) {
        MgShard& S = mg->sh[s];
        HIP_TRY(hipSetDevice(S.device));
        CtxLock lock(S.ctx->mu);
        ACX_TRY(mg_transpose(S.ctx, 2, d_blocks[s], S.io, q, P, nullptr));                    // [q][P] -> [P][q]
        HIP_TRY(hipMemcpy2DAsync(host + (uint64_t)s * q, stride * 32, S.io, q * 32, q * 32, P, hipMemcpyDeviceToHost, S.ctx->stream));
    }
    for (auto& S : mg->sh) {
        HIP_TRY(hipSetDevice(S.device));
        HIP_TRY(hipStreamSynchronize(S.ctx->stream));
    }
    return ACX_OK;
}

int mg_push_natural(acx_mgpu* mg, const acx_fr* host, uint64_t P, uint64_t q, uint64_t stride, uint4* const* d_blocks) {
    return mg_per_shard_threads(mg, [&](uint32_t s) -> int {
        MgShard& S = mg->sh[s];
        HIP_TRY(hipSetDevice(S.device));
        CtxLock lock(S.ctx->mu);
        HIP_TRY(hipMemsetAsync(S.d_res + 2, 0, 4, S.ctx->stream));
        HIP_TRY(hipMemcpy2DAsync(S.io, q * 32, host + (uint64_t)s * q, stride * 32, q * 32, P, hipMemcpyHostToDevice, S.ctx->stream));
        return mg_transpose(S.ctx, 1, S.io, d_blocks[s], P, q, (uint32_t*)(S.d_res + 2));     // [P][q] -> [q][P]
    });
}

int mg_check_canonical(acx_mgpu* mg) {
    for (auto& S : mg->sh) {
        HIP_TRY(hipSetDevice(S.device));
        uint32_t flag = 0;
        HIP_TRY(hipMemcpyAsync(&flag, S.d_res + 2, 4, hipMemcpyDeviceToHost, S.ctx->stream));
        HIP_TRY(hipStreamSynchronize(S.ctx->stream));
        if (flag) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    }
    return ACX_OK;
}

// the block-cyclic rows of one shard in ASCENDING order (local row j = [k2][kl]: runs of R/W consecutive rows, one run out of
// every R), gathered from the caller's CSR (rows >= n: empty)
struct ShardRows {
    std::vector<uint32_t> rowptr, col;
    std::vector<acx_fr> val;
};
void mg_gather_rows(const acx_csr& M, uint64_t n, uint32_t log_n, uint32_t log_r, uint32_t W, uint32_t g, ShardRows& out) {
    const uint64_t R = 1ull << log_r, C = 1ull << (log_n - log_r), rw = R / W, L = rw * C;
    const uint32_t log_rw = log_r - mg_log2(W);
    auto global_row = [&](uint64_t j) { return (uint64_t)g * rw + (j & (rw - 1)) + ((j >> log_rw) << log_r); };     // ascending
    out.rowptr.assign(L + 1, 0);
    uint64_t nnz = 0;
    for (uint64_t j = 0; j < L; ++j) {
        const uint64_t row = global_row(j);
        if (row < n) nnz += M.rowptr[row + 1] - M.rowptr[row];
        out.rowptr[j + 1] = (uint32_t)nnz;
    }
    out.col.resize(nnz);
    out.val.resize(nnz);
    // the copies run on a few worker threads per shard (the shards themselves are gathered concurrently, one thread each)
    parallel_ranges(L, std::min(8u, host_threads(L, 1 << 16)), [&](unsigned, uint64_t jb, uint64_t je) {
        for (uint64_t j = jb; j < je; ++j) {
            const uint64_t row = global_row(j);
            if (row >= n) continue;
            const uint32_t e0 = M.rowptr[row], len = M.rowptr[row + 1] - e0;
            if (len == 0) continue;
            std::memcpy(&out.col[out.rowptr[j]], M.col + e0, (size_t)len * 4);
            std::memcpy(&out.val[out.rowptr[j]], M.val + e0, (size_t)len * 32);
        }
    });
}

void mg_free_r1cs(acx_mgpu_r1cs* mr) {
    if (!mr) return;
    acx_mgpu* mg = mr->mg;
    if (mr->whole) acx_r1cs_destroy(mr->whole);
    for (size_t s = 0; s < mr->part.size(); ++s) {
        auto& p = mr->part[s];
        if (p.slab) acx_r1cs_destroy(p.slab);                         // synchronises that device
        if (p.cyc) acx_r1cs_destroy(p.cyc);
        if (p.full) acx_r1cs_destroy(p.full);
        if (p.cols) acx_r1cs_destroy(p.cols);
        (void)hipSetDevice(mg->sh[s].device);
        if (p.d_w) (void)hipFree(p.d_w);
        if (p.vec) (void)hipFree(p.vec);
        if (p.ring) (void)hipFree(p.ring);
        if (p.hscale) (void)hipFree(p.hscale);
    }
    delete mr;
}

// contiguous slabs balanced by nnz(A) + nnz(B) + nnz(C) + 1 per row (Split gates make 257-row bursts of uneven cost,
// test/Test/Circuit/Arithmetic.hs:123): W + 1 boundaries
std::vector<uint64_t> mg_slab_bounds(const acx_csr* const mats[3], uint64_t n, uint32_t W) {
    auto cost = [&](uint64_t i) { return (uint64_t)mats[0]->rowptr[i] + mats[1]->rowptr[i] + mats[2]->rowptr[i] + i; };
    const uint64_t total = cost(n);
    std::vector<uint64_t> b(W + 1, n);
    b[0] = 0;
    for (uint32_t r = 1; r < W; ++r) {
        const uint64_t want = total / W * r;
        uint64_t lo = b[r - 1], hi = n;
        while (lo < hi) { const uint64_t mid = (lo + hi) / 2; if (cost(mid) < want) lo = mid + 1; else hi = mid; }
        b[r] = lo;
    }
    return b;
}

int mg_load(acx_mgpu* mg, uint64_t n, uint64_t m, const acx_csr* const mats[3], uint32_t flags, acx_mgpu_r1cs** out) {
    if (m == 0 || m >= 0xffffffffull || n >= 0xffffffffull) return fail(ACX_ERR_TOO_LARGE, "n or m out of range");
    if (flags & ~(uint32_t)ACX_MGPU_VERIFY_ONLY) return fail(ACX_ERR_INVALID_ARG, "unknown load flag");
    const uint32_t log_n = ceil_log2(std::max<uint64_t>(n, 1));
    if ((int)log_n > mg->sh[0].ctx->hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "n exceeds 2^two_adicity");
    for (int k = 0; k < 3; ++k) {
        if (!mats[k] || !mats[k]->rowptr) return fail(ACX_ERR_INVALID_ARG, "null CSR");
        if (mats[k]->rowptr[0] != 0) return fail(ACX_ERR_INVALID_ARG, "rowptr[0] != 0");
        std::atomic<bool> bad{false};                          // checked before any row is gathered: the gathers trust the row pointers
        parallel_ranges(n, host_threads(n, 1 << 18), [&](unsigned, uint64_t b, uint64_t e) {
            for (uint64_t i = b; i < e; ++i)
                if (mats[k]->rowptr[i + 1] < mats[k]->rowptr[i]) { bad = true; return; }
        });
        if (bad) return fail(ACX_ERR_INVALID_ARG, "rowptr not monotone");
        if (mats[k]->rowptr[n] && (!mats[k]->col || !mats[k]->val)) return fail(ACX_ERR_INVALID_ARG, "null CSR arrays");
    }
    std::unique_ptr<acx_mgpu_r1cs> mr(new acx_mgpu_r1cs());
    mr->mg = mg; mr->n = n; mr->m = m; mr->log_n = log_n;
    const uint32_t W = mg->W;
    // One shard is "sharded" too when the size allows the four-step transform (its exchange is RCCL's all-to-all with itself):
    // the same code path at every n_devices.  Several shards split any system at or above the threshold; the block-cyclic
    // copy for h(x) exists where the transforms can be distributed (mg_can_distribute).
    const bool can_h = mg_can_distribute(W, log_n);
    mr->sharded = log_n >= mg->min_log_n && (W > 1 || can_h);
    if (!mr->sharded) {
        ACX_TRY(r1cs_from_host(mg->sh[0].ctx, n, m, mats, &mr->whole));
        *out = mr.release();
        return ACX_OK;
    }
    mr->verify_only = (flags & ACX_MGPU_VERIFY_ONLY) != 0;
    mr->has_cyclic = can_h && !mr->verify_only;
    mr->log_r = log_n / 2;
    mr->part.resize(W);
    const uint64_t L = (1ull << log_n) / W;
    const std::vector<uint64_t> bounds = mg_slab_bounds(mats, n, W);
    const int rc = mg_per_shard_threads(mg, [&](uint32_t s) -> int {
        auto& P = mr->part[s];
        HIP_TRY(hipSetDevice(mg->sh[s].device));
        {   // the slab: views into the caller's arrays, row pointers rebased
            const uint64_t b0 = bounds[s], b1 = bounds[s + 1];
            std::vector<uint32_t> rp[3];
            acx_csr views[3];
            const acx_csr* mp[3];
            for (int k = 0; k < 3; ++k) {
                const uint32_t e0 = mats[k]->rowptr[b0];
                rp[k].resize(b1 - b0 + 1);
                for (uint64_t i = b0; i <= b1; ++i) rp[k][i - b0] = mats[k]->rowptr[i] - e0;
                views[k] = acx_csr{rp[k].data(), mats[k]->col ? mats[k]->col + e0 : nullptr, mats[k]->val ? mats[k]->val + e0 : nullptr};
                mp[k] = &views[k];
            }
            P.row0 = b0;
            ACX_TRY(r1cs_from_host(mg->sh[s].ctx, b1 - b0, m, mp, &P.slab));
        }
        if (mr->has_cyclic) {
            ShardRows rows[3];
            acx_csr views[3];
            const acx_csr* mp[3];
            for (int k = 0; k < 3; ++k) {
                mg_gather_rows(*mats[k], n, log_n, mr->log_r, W, s, rows[k]);
                views[k] = acx_csr{rows[k].rowptr.data(), rows[k].col.data(), rows[k].val.data()};
                mp[k] = &views[k];
            }
            ACX_TRY(r1cs_from_host(mg->sh[s].ctx, L, m, mp, &P.cyc));
        }
        HIP_TRY(hipMalloc((void**)&P.d_w, m * 32));
        if (mr->has_cyclic && (int)log_n + 1 <= mg->sh[s].ctx->hf.two_adicity()) {
            const HostField& hf = mg->sh[s].ctx->hf;
            const H256 zinv = hf.inv(hf.sub(hf.pow_u64(hf.generator(), 1ull << log_n), hf.one()));
            const H256 pair[2] = {hf.to_dev_word(zinv), hf.to_dev_word(hf.sub(hf.zero(), zinv))};
            HIP_TRY(hipMalloc((void**)&P.hscale, 64));
            HIP_TRY(hipMemcpy(P.hscale, pair, 64, hipMemcpyHostToDevice));
        }
        HIP_TRY(hipMalloc((void**)&P.ring, 4 * 2 * kMgRing * 8));
        std::vector<unsigned long long> init(4 * 2 * kMgRing);
        for (uint32_t i = 0; i < 4 * kMgRing; ++i) { init[2 * i] = 0; init[2 * i + 1] = ~0ull; }
        HIP_TRY(hipMemcpy(P.ring, init.data(), init.size() * 8, hipMemcpyHostToDevice));
        return ACX_OK;
    });
    if (rc != ACX_OK) { mg_free_r1cs(mr.release()); return rc; }
    *out = mr.release();
    return ACX_OK;
}

// A copy of the WHOLE system on shard 0 (every_shard = false is the only use left): acx_mgpu_qap_h of a transform size the
// distributed four-step form does not cover answers from one device.  The slabs are read back from the devices (canonical CSR,
// acx_r1cs_export), joined on the host and loaded.
int mg_ensure_replicas(acx_mgpu_r1cs* mr, bool every_shard = true) {
    acx_mgpu* mg = mr->mg;
    const uint32_t W = mg->W;
    if (!mr->sharded) return ACX_OK;
    bool missing = false;
    for (uint32_t s = 0; s < (every_shard ? W : 1u); ++s) missing = missing || !mr->part[s].full;
    if (!missing) return ACX_OK;
    const uint64_t n = mr->n;
    std::vector<uint64_t> nnz0[3];                  // entry offset of every slab in the joined matrix
    for (int k = 0; k < 3; ++k) nnz0[k].assign(W + 1, 0);
    for (uint32_t s = 0; s < W; ++s) {
        uint64_t z[3] = {0, 0, 0};
        ACX_TRY(acx_r1cs_dims(mr->part[s].slab, nullptr, nullptr, nullptr, z));
        for (int k = 0; k < 3; ++k) nnz0[k][s + 1] = nnz0[k][s] + z[k];
    }
    std::vector<uint32_t> rowptr[3], col[3];
    std::vector<acx_fr> val[3];
    for (int k = 0; k < 3; ++k) {
        if (nnz0[k][W] >= 0xffffffffull) return fail(ACX_ERR_TOO_LARGE, "matrix has 2^32 entries or more");
        rowptr[k].assign(n + 1, 0);
        col[k].resize(nnz0[k][W]);
        val[k].resize(nnz0[k][W]);
    }
    ACX_TRY(mg_per_shard_threads(mg, [&](uint32_t s) -> int {
        const auto& P = mr->part[s];
        uint64_t rows = 0;
        ACX_TRY(acx_r1cs_dims(P.slab, &rows, nullptr, nullptr, nullptr));
        std::vector<uint32_t> rp(rows + 1);
        for (int k = 0; k < 3; ++k) {
            const uint64_t e0 = nnz0[k][s];
            ACX_TRY(acx_r1cs_export(P.slab, k, rp.data(), col[k].data() + e0, val[k].data() + e0));
            for (uint64_t i = 1; i <= rows; ++i) rowptr[k][P.row0 + i] = (uint32_t)(e0 + rp[i]);   // slabs are disjoint row ranges
        }
        return ACX_OK;
    }));
    acx_csr views[3];
    const acx_csr* mp[3];
    for (int k = 0; k < 3; ++k) {
        views[k] = acx_csr{rowptr[k].data(), col[k].data(), val[k].data()};
        mp[k] = &views[k];
    }
    return mg_per_shard_threads(mg, [&](uint32_t s) -> int {          // a shard that fails keeps no copy; the others keep theirs
        if (mr->part[s].full || (!every_shard && s != 0)) return ACX_OK;
        HIP_TRY(hipSetDevice(mg->sh[s].device));
        return r1cs_from_host(mg->sh[s].ctx, n, mr->m, mp, &mr->part[s].full);
    });
}

// Per-wire polynomials (`createPolynomialsFFT`, src/QAP.hs:512-525) shard by WIRE with no communication, and a
// wire's interpolation needs its own COLUMN of every row, nothing else.  So each shard holds the column view of its wires only:
// wire w belongs to shard (w / kMgColBlock) mod W (block-cyclic: any request of a few hundred consecutive wires spreads over all
// devices), numbered locally (w / (kMgColBlock W)) * kMgColBlock + w mod kMgColBlock.  All shards together hold every entry
// ONCE (40 bytes each: row, column, value) -- the first version gave every device a copy of the whole system.  Built on the
// first call, one matrix at a time: every shard's row slab is read back (canonical CSR), a counting sort by column makes the
// global column view on the host, every shard takes its blocks.
int mg_ensure_col_slices(acx_mgpu_r1cs* mr) {
    acx_mgpu* mg = mr->mg;
    const uint32_t W = mg->W;
    if (!mr->sharded || mr->part[0].cols) return ACX_OK;
    const uint64_t m = mr->m, B = kMgColBlock;
    auto owner = [&](uint64_t w) { return (uint32_t)((w / B) % W); };
    auto local = [&](uint64_t w) { return (w / (B * W)) * B + w % B; };
    std::vector<uint64_t> m_local(W, 0);
    for (uint64_t j = 0; j * B < m; ++j) m_local[j % W] += std::min<uint64_t>(B, m - j * B);
    std::vector<std::array<HostCsc, 3>> slices(W);
    for (int k = 0; k < 3; ++k) {
        // read the slabs back: rows [row0, row0 + rows) of matrix k per shard
        std::vector<std::vector<uint32_t>> rp(W), cl(W);
        std::vector<std::vector<acx_fr>> vl(W);
        ACX_TRY(mg_per_shard_threads(mg, [&](uint32_t s) -> int {
            const auto& P = mr->part[s];
            uint64_t rows = 0, z[3] = {0, 0, 0};
            ACX_TRY(acx_r1cs_dims(P.slab, &rows, nullptr, nullptr, z));
            rp[s].resize(rows + 1);
            cl[s].resize(z[k]);
            vl[s].resize(z[k]);
            return acx_r1cs_export(P.slab, k, rp[s].data(), cl[s].data(), vl[s].data());
        }));
        // counting sort by column over all slabs -> global colptr; then every shard's blocks in local numbering
        std::vector<uint64_t> colptr(m + 1, 0);
        for (uint32_t s = 0; s < W; ++s)
            for (uint32_t c : cl[s]) ++colptr[(uint64_t)c + 1];
        for (uint64_t w = 0; w < m; ++w) colptr[w + 1] += colptr[w];
        for (uint32_t s = 0; s < W; ++s) {                           // local colptr of every shard
            HostCsc& H = slices[s][k];
            H.colptr.assign(m_local[s] + 1, 0);
        }
        for (uint64_t w = 0; w < m; ++w) slices[owner(w)][k].colptr[local(w) + 1] = (uint32_t)(colptr[w + 1] - colptr[w]);
        for (uint32_t s = 0; s < W; ++s) {
            HostCsc& H = slices[s][k];
            uint64_t acc = 0;
            for (uint64_t i = 0; i < m_local[s]; ++i) { acc += H.colptr[i + 1]; if (acc >= 0xffffffffull) return fail(ACX_ERR_TOO_LARGE, "column slice has 2^32 entries or more"); H.colptr[i + 1] = (uint32_t)acc; }
            H.rowidx.resize(acc); H.colid.resize(acc); H.val.resize(acc);
        }
        std::vector<uint32_t> cursor(m, 0);
        for (uint32_t s = 0; s < W; ++s) {                           // slabs in shard order = ascending global rows
            const uint64_t row0 = mr->part[s].row0;
            const uint64_t rows = rp[s].size() - 1;
            for (uint64_t i = 0; i < rows; ++i)
                for (uint32_t e = rp[s][i]; e < rp[s][i + 1]; ++e) {
                    const uint64_t w = cl[s][e];
                    HostCsc& H = slices[owner(w)][k];
                    const uint64_t lw = local(w), dst = (uint64_t)H.colptr[lw] + cursor[w]++;
                    H.rowidx[dst] = (uint32_t)(row0 + i);
                    H.colid[dst] = (uint32_t)lw;
                    H.val[dst] = vl[s][e];
                }
        }
    }
    const int rc = mg_per_shard_threads(mg, [&](uint32_t s) -> int {
        return r1cs_column_slice_from_host(mg->sh[s].ctx, mr->n, mr->log_n, m_local[s], slices[s].data(), &mr->part[s].cols);
    });
    if (rc != ACX_OK)                                               // all or none: a retry starts clean
        for (uint32_t s = 0; s < W; ++s)
            if (mr->part[s].cols) { acx_r1cs_destroy(mr->part[s].cols); mr->part[s].cols = nullptr; }
    return rc;
}

// verificationWitnessZk over the shards on the resident witness; h stays on the devices in COLS ownership.
// The whole pipeline of ONE shard -- residual launch, six transforms (twelve local steps, six exchanges), the elementwise tail --
// is issued by that shard's own thread (mg_qap_h_issue_shard); the calling thread then waits once, for the verdict.
struct MgHArgs {
    const H256* dl;
    bool zk, fusedh;
    H256 g, ginv, zinv, mzinv;
};

int mg_qap_h_issue_shard(acx_mgpu_r1cs* mr, uint32_t s, const MgHArgs& A) {
    acx_mgpu* mg = mr->mg;
    const uint32_t W = mg->W;
    const uint64_t N = 1ull << mr->log_n, L = N / W;
    MgShard& S = mg->sh[s];
    const HostField& hf = S.ctx->hf;
    HIP_TRY(hipSetDevice(S.device));
    uint4* v = mr->part[s].vec;
    auto at = [&](uint64_t off) { return v + 2 * off * L; };
    // vec: dots k at k L (ROWS), coefficients k at (3 + k) L (COLS), pointwise product at 6 L (ROWS), h at 7 L (COLS)
    ACX_TRY(mg_residual_enqueue_shard(mr, s, true, A.fusedh));      // the verdict is fetched after the whole pipeline has been issued: one wait
    MgNtt nt(mg, mr->log_n, mr->log_r);
    // Software pipeline over the three vectors (qap_h_dev_locked's sequence, sharded): vector k's exchange runs on the
    // exchange stream under vector k+1's local step, and a vector's coset transform starts as soon as its inverse one is
    // complete -- of the six all-to-alls only the last has no local work to hide behind.
    // Without the zero-knowledge terms nobody needs the plain coefficients of L and R: their coset factor g^i rides on the
    // closing multiplication of their INVERSE transform (an inverse coset transform with shift 1/g multiplies by g^i: +9 us on
    // a step that otherwise closes with a plain reduction) instead of on the load of the forward one (-35 us: one product per
    // element less), as in the single-GPU pipeline (qap_h_dev_locked).  O stays in plain coefficient form.
    static const bool on_forward = [] { const char* e = std::getenv("ACX_MGPU_COSET_ON_FORWARD"); return e && std::atoi(e) != 0; }();   // development A/B: round 3's sequence
    const bool fold = A.fusedh && !on_forward;
    const H256* up = fold ? &A.ginv : nullptr;                      // shift of the inverse transforms of L and R
    const H256* fw = fold ? nullptr : &A.g;                         // shift of their forward transforms
    for (int k = 0; k < 3; ++k) ACX_TRY(nt.begin(s, k, at(k), 1, k < 2 ? up : nullptr, true));              // dots: ascending row order
    for (int k = 0; k < 3; ++k) {
        ACX_TRY(nt.finish(s, k, at(3 + k), 1, k < 2 ? up : nullptr));
        if (k < 2) ACX_TRY(nt.begin(s, k, at(3 + k), 0, fw));
    }
    for (int k = 0; k < 2; ++k) ACX_TRY(nt.finish(s, k, at(k), 0, fw));
    if (A.fusedh) {
        // without the zero-knowledge terms 1/z and -1/z ride on the stored dots, the last transform takes (L/z) * R as its first
        // step loads the points and adds -O/z behind its closing step (qap_h_dev_locked's fused form, sharded)
        ACX_TRY(nt.begin(s, 0, at(0), 1, &A.g, false, at(1)));
        return nt.finish(s, 0, at(7), 1, &A.g, at(5));
    }
    {
        CtxLock lock(S.ctx->mu);
        DISPATCH_FIELD(S.ctx, hipLaunchKernelGGL((k_pointwise_h<F>), dim3(grid_for(S.ctx, L)), dim3(kBlock), 0, S.ctx->stream, (const uint4*)v,
                                                 (const uint4*)(v + 2 * L), (const uint4*)nullptr, v + 2 * 6 * L, L, dev_arg(hf, A.zinv), 0u));
        HIP_TRY(hipGetLastError());
    }
    ACX_TRY(nt.begin(s, 0, at(6), 1, &A.g));
    ACX_TRY(nt.finish(s, 0, at(7), 1, &A.g));
    CtxLock lock(S.ctx->mu);
    uint4 *h = at(7), *L0 = at(3), *R0 = at(4), *O0 = at(5);
    if (A.zk) {
        // (L0+d1 T)(R0+d2 T) - (O0+d3 T) = T (h0 + d1 R0 + d2 L0 + d1 d2 T - d3), T = x^N - 1 (src/QAP.hs:315-323): the
        // elementwise part is layout agnostic (h, L0, R0, O0 share the COLS ownership); coefficient 0 lives on shard 0
        // at local index 0 and coefficient N (= d1 d2) is appended by the fetch
        const H256 d12 = hf.mul(A.dl[0], A.dl[1]);
        DISPATCH_FIELD(S.ctx, {
            hipLaunchKernelGGL((k_axpy3<F>), dim3(grid_for(S.ctx, L)), dim3(kBlock), 0, S.ctx->stream, h, (const uint4*)R0, (const uint4*)L0,
                               (const uint4*)O0, L, dev_arg(hf, A.dl[0]), dev_arg(hf, A.dl[1]), dev_arg(hf, A.mzinv));
            if (s == 0) hipLaunchKernelGGL((k_h_fix<F>), dim3(1), dim3(64), 0, S.ctx->stream, h, ~(u64)0, dev_arg(hf, hf.add(d12, A.dl[2])),
                                           dev_arg(hf, hf.zero()));
        });
    } else {
        DISPATCH_FIELD(S.ctx, hipLaunchKernelGGL((k_axpy3<F>), dim3(grid_for(S.ctx, L)), dim3(kBlock), 0, S.ctx->stream, h, (const uint4*)nullptr,
                                                 (const uint4*)nullptr, (const uint4*)O0, L, dev_arg(hf, A.mzinv), dev_arg(hf, A.mzinv), dev_arg(hf, A.mzinv)));
    }
    HIP_TRY(hipGetLastError());
    return ACX_OK;
}

int mg_qap_h_resident(acx_mgpu_r1cs* mr, const H256* dl, bool* ok) {
    acx_mgpu* mg = mr->mg;
    const uint32_t W = mg->W;
    const HostField& hf = mg->sh[0].ctx->hf;
    if (!mr->has_cyclic)
        return fail(ACX_ERR_UNSUPPORTED, "no block-cyclic copy of this system: loaded with ACX_MGPU_VERIFY_ONLY, or N outside 2^10 .. 2^24 / 2 W > sqrt(N) (acx_mgpu_qap_h then answers from one device)");
    if ((int)mr->log_n + 1 > hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "coset needs log_n + 1 <= two-adicity");
    const uint64_t N = 1ull << mr->log_n, L = N / W;
    ACX_TRY(mg_ensure_slots(mg, L));
    for (uint32_t s = 0; s < W; ++s)
        if (!mr->part[s].vec) {
            HIP_TRY(hipSetDevice(mg->sh[s].device));
            HIP_TRY(hipMalloc((void**)&mr->part[s].vec, 8 * L * 32));
        }
    mr->h_valid = false;
    MgClock clock(mg);
    MgHArgs A;
    A.dl = dl;
    A.zk = dl && !(dl[0].is_zero() && dl[1].is_zero() && dl[2].is_zero());
    A.fusedh = !A.zk && mr->part[0].hscale != nullptr;
    A.g = hf.generator();
    A.ginv = hf.inv(A.g);
    A.zinv = hf.inv(hf.sub(hf.pow_u64(A.g, N), hf.one()));
    A.mzinv = hf.sub(hf.zero(), A.zinv);
    ACX_TRY(mg_per_shard_threads(mg, [&](uint32_t s) -> int { return mg_qap_h_issue_shard(mr, s, A); }));
    uint64_t n_bad = 0, first = 0;
    bool noncanon = false;
    clock.issued();
    ACX_TRY(mg_residual_fetch(mr, false, &n_bad, &first, &noncanon));
    if (noncanon) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    mr->h_top = A.zk ? hf.mul(dl[0], dl[1]) : hf.zero();
    mr->h_valid = true;
    *ok = n_bad == 0;
    return ACX_OK;
}

int mg_sync(acx_mgpu* mg) {
    for (auto& S : mg->sh) {
        HIP_TRY(hipSetDevice(S.device));
        HIP_TRY(hipStreamSynchronize(S.ctx->stream));
        HIP_TRY(hipStreamSynchronize(S.xstream));
    }
    return ACX_OK;
}

}  // namespace

extern "C" {

void acx_mgpu_destroy(acx_mgpu* mg) {
    if (!mg) return;
    DevGuard dg;
    if (mg->pool) mg->pool->shutdown();
    for (auto& S : mg->sh) {
        if (!S.ctx) continue;                                      // creation stopped before this shard: nothing to release
        (void)hipSetDevice(S.device);
        (void)hipDeviceSynchronize();
        if (S.comm && mg->api) (void)mg->api->CommDestroy(S.comm);
        for (auto& sl : S.slot) {
            if (sl.send) (void)hipFree(sl.send);
            if (sl.recv) (void)hipFree(sl.recv);
            if (sl.sent) (void)hipEventDestroy(sl.sent);
            if (sl.got) (void)hipEventDestroy(sl.got);
            if (sl.used) (void)hipEventDestroy(sl.used);
        }
        if (S.io) (void)hipFree(S.io);
        if (S.w_ready) (void)hipEventDestroy(S.w_ready);
        if (S.w_read) (void)hipEventDestroy(S.w_read);
        if (S.d_res) (void)hipFree(S.d_res);
        if (S.xstream) (void)hipStreamDestroy(S.xstream);
        if (S.ctx) acx_ctx_destroy(S.ctx);
    }
    delete mg;
}

int acx_mgpu_create(int field, const int* device_ids, uint32_t n_devices, acx_mgpu** out) {
    if (!out || !device_ids || n_devices == 0) return fail(ACX_ERR_INVALID_ARG, "null / empty device list");
    if (n_devices > 64 || (n_devices & (n_devices - 1))) return fail(ACX_ERR_INVALID_ARG, "n_devices must be a power of two (<= 64): the shards split both factors of N");
    return guarded([&]() -> int {
        DevGuard dg;
        std::unique_ptr<acx_mgpu, void (*)(acx_mgpu*)> mg(new acx_mgpu(), acx_mgpu_destroy);
        mg->field = field;
        mg->W = n_devices;
        mg->sh.resize(n_devices);
        bool distinct = true;
        for (uint32_t i = 0; i < n_devices; ++i)
            for (uint32_t j = 0; j < i; ++j) distinct = distinct && device_ids[i] != device_ids[j];
        const char* tr = std::getenv("ACX_MGPU_TRANSPORT");
        const bool want_peer = tr && std::string(tr) == "peer";
        if (tr && !want_peer && std::string(tr) != "rccl") return fail(ACX_ERR_INVALID_ARG, "ACX_MGPU_TRANSPORT must be rccl or peer");
        if (tr && !want_peer && !distinct) return fail(ACX_ERR_INVALID_ARG, "RCCL needs distinct devices (a device list with repeats uses peer copies)");
        mg->rccl = distinct && !want_peer;
        for (uint32_t i = 0; i < n_devices; ++i) {
            MgShard& S = mg->sh[i];
            ACX_TRY(acx_ctx_create(field, device_ids[i], &S.ctx));                       // validates the device (gfx950) and the field
            S.device = device_ids[i];
            HIP_TRY(hipSetDevice(S.device));
            HIP_TRY(hipStreamCreateWithFlags(&S.xstream, hipStreamNonBlocking));
            HIP_TRY(hipMalloc((void**)&S.d_res, 64));
            HIP_TRY(hipMemset(S.d_res, 0, 64));
            for (auto& sl : S.slot) {
                HIP_TRY(hipEventCreateWithFlags(&sl.sent, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&sl.got, hipEventDisableTiming));
                HIP_TRY(hipEventCreateWithFlags(&sl.used, hipEventDisableTiming));
            }
            HIP_TRY(hipEventCreateWithFlags(&S.w_ready, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&S.w_read, hipEventDisableTiming));
        }
        if (const char* wm = std::getenv("ACX_MGPU_WITNESS")) {
            const std::string m(wm);
            if (m == "broadcast") mg->witness_mode = 0;
            else if (m == "copies") mg->witness_mode = 1;
            else if (m == "pinned") mg->witness_mode = 2;
            else return fail(ACX_ERR_INVALID_ARG, "ACX_MGPU_WITNESS must be broadcast, copies or pinned");
        }
        if (n_devices > 1) {                                        // one issuing thread per shard for the life of the handle
            mg->pool.reset(new MgPool());
            mg->pool->start(n_devices, std::vector<int>(device_ids, device_ids + n_devices));
        }
        if (mg->rccl) {
            std::string why;
            mg->api = rccl_api(why);
            // no usable RCCL on this machine: the peer-copy transport carries the same events, buffers and results (an explicit
            // ACX_MGPU_TRANSPORT=rccl is an error instead: the caller asked for the collectives)
            if (!mg->api && tr) return fail(ACX_ERR_UNSUPPORTED, why);
            if (!mg->api) mg->rccl = false;
        }
        if (mg->rccl) {
            std::vector<ncclComm_t> comms(n_devices);
            NCCL_TRY(mg, mg->api->CommInitAll(comms.data(), (int)n_devices, device_ids));
            for (uint32_t i = 0; i < n_devices; ++i) mg->sh[i].comm = comms[i];
        } else {
            for (uint32_t i = 0; i < n_devices; ++i)
                for (uint32_t j = 0; j < n_devices; ++j) {
                    if (device_ids[i] == device_ids[j]) continue;
                    int can = 0;
                    HIP_TRY(hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]));
                    if (!can) continue;                                                     // hipMemcpyPeerAsync then stages through the host
                    HIP_TRY(hipSetDevice(device_ids[i]));
                    const hipError_t e = hipDeviceEnablePeerAccess(device_ids[j], 0);
                    if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) HIP_TRY(e);
                    (void)hipGetLastError();
                }
        }
        *out = mg.release();
        return ACX_OK;
    });
}

int acx_mgpu_info(const acx_mgpu* mg, uint32_t* n_devices, int* transport, uint32_t* shard_threshold_log_n) {
    if (!mg) return fail(ACX_ERR_INVALID_ARG, "null handle");
    if (n_devices) *n_devices = mg->W;
    if (transport) *transport = mg->rccl ? ACX_MGPU_RCCL : ACX_MGPU_PEER_COPY;
    if (shard_threshold_log_n) *shard_threshold_log_n = mg->min_log_n;
    return ACX_OK;
}

acx_ctx* acx_mgpu_ctx(acx_mgpu* mg, uint32_t shard) { return (mg && shard < mg->W) ? mg->sh[shard].ctx : nullptr; }

// development aid (not in include/acx.h): {issue seconds, total seconds} of the last verify / h(x) call on the handle
int acx_mgpu_debug_times(acx_mgpu* mg, double out[2]) {
    if (!mg || !out) return ACX_ERR_INVALID_ARG;
    out[0] = mg->last_issue_s; out[1] = mg->last_total_s;
    return ACX_OK;
}

int acx_mgpu_set_shard_threshold(acx_mgpu* mg, uint32_t log_n) {
    if (!mg) return fail(ACX_ERR_INVALID_ARG, "null handle");
    std::lock_guard<std::mutex> g(mg->mu);
    mg->min_log_n = std::max<uint32_t>(10, log_n);
    return ACX_OK;
}

int acx_mgpu_set_root(acx_mgpu* mg, uint32_t two_adicity, const acx_fr* omega) {
    return guarded([&]() -> int {
        if (!mg) return fail(ACX_ERR_INVALID_ARG, "null handle");
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        for (auto& S : mg->sh) ACX_TRY(acx_ctx_set_root(S.ctx, two_adicity, omega));
        return ACX_OK;
    });
}

int acx_mgpu_sync(acx_mgpu* mg) {
    return guarded([&]() -> int {
        if (!mg) return fail(ACX_ERR_INVALID_ARG, "null handle");
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        return mg_sync(mg);
    });
}

int acx_mgpu_r1cs_load(acx_mgpu* mg, uint64_t n, uint64_t m, const acx_csr* A, const acx_csr* B, const acx_csr* C, uint32_t flags,
                       acx_mgpu_r1cs** out) {
    ACX_RANGE();
    if (!mg || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    const acx_csr* mats[3] = {A, B, C};
    std::lock_guard<std::mutex> g(mg->mu);
    DevGuard dg;
    return guarded([&]() -> int { return mg_load(mg, n, m, mats, flags, out); });
}

int acx_mgpu_circuit_to_r1cs(acx_mgpu* mg, const acx_circuit* c, const acx_fr* roots, uint64_t n_roots, uint32_t flags,
                             acx_mgpu_r1cs** out) {
    ACX_RANGE();
    if (!mg || !c || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (c->field != mg->field) return fail(ACX_ERR_INVALID_ARG, "circuit and context are over different fields");
    std::lock_guard<std::mutex> g(mg->mu);
    DevGuard dg;
    return guarded([&]() -> int {
        const HostCircuit& hc = c->hc;
        const uint64_t n = hc.n_rows();
        const uint32_t log_n = ceil_log2(std::max<uint64_t>(n, 1));
        const bool shard = log_n >= mg->min_log_n && (mg->W > 1 || mg_can_distribute(mg->W, log_n));
        if (!shard) {                                                   // small system: shard 0 holds it whole, with its evaluation plan
            std::unique_ptr<acx_mgpu_r1cs> mr(new acx_mgpu_r1cs());
            mr->mg = mg; mr->n = n; mr->m = hc.m(); mr->log_n = log_n;
            ACX_TRY(circuit_to_r1cs_impl(mg->sh[0].ctx, c, roots, n_roots, &mr->whole));
            *out = mr.release();
            return ACX_OK;
        }
        std::vector<uint64_t> order;
        ACX_TRY(root_order(hc, roots, n_roots, order));
        acx_csr views[3];
        HostCsr P[3];
        const acx_csr* mats[3];
        for (int k = 0; k < 3; ++k) {
            const HostCsr* src = &c->rows[k];
            if (!order.empty()) { permute_rows(*src, order, P[k]); src = &P[k]; }
            views[k] = acx_csr{src->rowptr.data(), src->col.data(), reinterpret_cast<const acx_fr*>(src->val.data())};
            mats[k] = &views[k];
        }
        return mg_load(mg, n, hc.m(), mats, flags, out);
    });
}

void acx_mgpu_r1cs_destroy(acx_mgpu_r1cs* mr) {
    if (!mr) return;
    std::lock_guard<std::mutex> g(mr->mg->mu);
    DevGuard dg;
    (void)mg_sync(mr->mg);
    mg_free_r1cs(mr);
}

int acx_mgpu_r1cs_dims(const acx_mgpu_r1cs* mr, uint64_t* n, uint64_t* m, uint32_t* log_n, uint32_t* n_shards) {
    if (!mr) return fail(ACX_ERR_INVALID_ARG, "null r1cs");
    if (n) *n = mr->n;
    if (m) *m = mr->m;
    if (log_n) *log_n = mr->log_n;
    if (n_shards) *n_shards = mr->sharded ? mr->mg->W : 1;
    return ACX_OK;
}

int acx_mgpu_witness_upload(acx_mgpu_r1cs* mr, const acx_fr* witness) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !witness) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "system is held whole on shard 0 (below the shard threshold): use the host-buffer calls");
        std::lock_guard<std::mutex> g(mr->mg->mu);
        DevGuard dg;
        ACX_TRY(mg_upload_witness(mr, witness));
        ACX_TRY(mg_sync(mr->mg));
        for (auto& S : mr->mg->sh) {
            uint32_t flag = 0;
            HIP_TRY(hipSetDevice(S.device));
            HIP_TRY(hipMemcpy(&flag, S.d_res + 2, 4, hipMemcpyDeviceToHost));
            if (flag) { mr->witness_resident = false; return fail(ACX_ERR_NONCANONICAL, "element >= p"); }
        }
        return ACX_OK;
    });
}

int acx_mgpu_r1cs_verify_resident(acx_mgpu_r1cs* mr, int* ok, uint64_t* n_bad, uint64_t* first_bad) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        std::lock_guard<std::mutex> g(mr->mg->mu);
        if (!mr->witness_resident) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        DevGuard dg;
        uint64_t bad = 0, first = ~0ull;
        bool noncanon = false;
        ACX_TRY(mg_residual(mr, false, first_bad != nullptr, &bad, &first, &noncanon));
        *ok = bad == 0;
        if (n_bad) *n_bad = bad;
        if (first_bad) *first_bad = first;
        return ACX_OK;
    });
}

int acx_mgpu_r1cs_verify(acx_mgpu_r1cs* mr, const acx_fr* witness, int* ok, uint64_t* n_bad, uint64_t* first_bad) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !witness || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return acx_r1cs_verify(mr->whole, witness, ok, n_bad, first_bad);
        std::lock_guard<std::mutex> g(mr->mg->mu);
        DevGuard dg;
        MgClock clock(mr->mg);
        ACX_TRY(mg_upload_witness(mr, witness));
        uint64_t bad = 0, first = ~0ull;
        bool noncanon = false;
        ACX_TRY(mg_residual(mr, false, first_bad != nullptr, &bad, &first, &noncanon, &clock));
        if (noncanon) { mr->witness_resident = false; return fail(ACX_ERR_NONCANONICAL, "element >= p"); }
        *ok = bad == 0;
        if (n_bad) *n_bad = bad;
        if (first_bad) *first_bad = first;
        return ACX_OK;
    });
}

// Throughput form of the resident check: enqueue accumulates the violated-row count of ONE verification into ring slot
// `slot` on every device and returns at once; verdicts reduces a range of slots with ONE collective and waits.
int acx_mgpu_r1cs_verify_enqueue(acx_mgpu_r1cs* mr, uint32_t slot) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || slot >= kMgRing) return fail(ACX_ERR_INVALID_ARG, "bad argument (slot < 16)");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        acx_mgpu* mg = mr->mg;
        std::lock_guard<std::mutex> g(mg->mu);
        if (!mr->witness_resident) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        DevGuard dg;
        MgClock clock(mg);
        return mg_per_shard_threads(mg, [&](uint32_t s) -> int {
            MgShard& S = mg->sh[s];
            HIP_TRY(hipSetDevice(S.device));
            CtxLock lock(S.ctx->mu);
            const auto& P = mr->part[s];
            return launch_residual(P.slab, P.d_w, P.row0, P.ring + 2 * slot, nullptr, nullptr, 0);
        });
    });
}

int acx_mgpu_r1cs_verdicts(acx_mgpu_r1cs* mr, uint32_t slot0, uint32_t count, uint64_t* n_bad) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !n_bad || count == 0 || slot0 + count > kMgRing) return fail(ACX_ERR_INVALID_ARG, "bad argument (slot0 + count <= 16)");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "system is held whole on shard 0");
        acx_mgpu* mg = mr->mg;
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        const uint32_t W = mg->W;
        std::vector<unsigned long long> host(2 * count), init(2 * count);
        std::vector<std::vector<unsigned long long>> per;
        // after the vectors the copies below read and write: every exit, an error return between the enqueues and the waits
        // included, first waits for all shards' streams
        struct DrainAll {
            acx_mgpu* mg;
            ~DrainAll() { for (uint32_t s = 0; s < mg->W; ++s) { (void)hipSetDevice(mg->sh[s].device); (void)hipStreamSynchronize(mg->sh[s].ctx->stream); } }
        } drain{mg};
        for (uint32_t i = 0; i < count; ++i) { init[2 * i] = 0; init[2 * i + 1] = ~0ull; }
        for (uint32_t i = 0; i < count; ++i) n_bad[i] = 0;
        if (mg->rccl) {
            // ONE all-reduce for the whole range ({n_bad, first_bad} pairs; the first_bad words are not meaningful after a
            // sum): into the second half of the ring buffer
            NCCL_TRY(mg, mg->api->GroupStart());
            for (uint32_t s = 0; s < W; ++s) {
                unsigned long long* ring = mr->part[s].ring;
                const ncclResult_t r = mg->api->AllReduce(ring + 2 * slot0, ring + 2 * kMgRing + 2 * slot0, 2 * count, ncclUint64, ncclSum,
                                                          mg->sh[s].comm, mg->sh[s].ctx->stream);
                if (r != ncclSuccess) { (void)mg->api->GroupEnd(); return fail(ACX_ERR_HIP, std::string("ncclAllReduce: ") + mg->api->GetErrorString(r)); }
            }
            NCCL_TRY(mg, mg->api->GroupEnd());
            for (uint32_t s = 0; s < W; ++s) {
                MgShard& S = mg->sh[s];
                HIP_TRY(hipSetDevice(S.device));
                if (s == 0) HIP_TRY(hipMemcpyAsync(host.data(), mr->part[0].ring + 2 * kMgRing + 2 * slot0, 16 * count, hipMemcpyDeviceToHost, S.ctx->stream));
                HIP_TRY(hipMemcpyAsync(mr->part[s].ring + 2 * slot0, init.data(), 16 * count, hipMemcpyHostToDevice, S.ctx->stream));
            }
            for (uint32_t s = 0; s < W; ++s) {
                HIP_TRY(hipSetDevice(mg->sh[s].device));
                HIP_TRY(hipStreamSynchronize(mg->sh[s].ctx->stream));
            }
            for (uint32_t i = 0; i < count; ++i) n_bad[i] = host[2 * i];
            return ACX_OK;
        }
        per.assign(W, std::vector<unsigned long long>(2 * count));
        for (uint32_t s = 0; s < W; ++s) {
            MgShard& S = mg->sh[s];
            HIP_TRY(hipSetDevice(S.device));
            HIP_TRY(hipMemcpyAsync(per[s].data(), mr->part[s].ring + 2 * slot0, 16 * count, hipMemcpyDeviceToHost, S.ctx->stream));
            HIP_TRY(hipMemcpyAsync(mr->part[s].ring + 2 * slot0, init.data(), 16 * count, hipMemcpyHostToDevice, S.ctx->stream));
        }
        for (uint32_t s = 0; s < W; ++s) {
            HIP_TRY(hipSetDevice(mg->sh[s].device));
            HIP_TRY(hipStreamSynchronize(mg->sh[s].ctx->stream));
            for (uint32_t i = 0; i < count; ++i) n_bad[i] += per[s][2 * i];
        }
        return ACX_OK;
    });
}

// `all (verifyAssignment qap) assignments` (test/Test/Circuit/Arithmetic.hs:209) over all devices in one call: every witness is
// replicated in turn (one copy per GPU, issued asynchronously by the shard's host thread into one of two witness buffers, so
// witness k+1 crosses PCIe while witness k is being checked), its check accumulates into a ring slot, and the verdicts of up to
// 16 witnesses are combined by ONE all-reduce.
int acx_mgpu_r1cs_verify_many(acx_mgpu_r1cs* mr, uint64_t count, const acx_fr* witnesses, uint8_t* ok, uint64_t* n_bad) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !ok || (count && !witnesses)) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (count == 0) return ACX_OK;
        if (!mr->sharded) return acx_r1cs_verify_many(mr->whole, count, witnesses, ok, n_bad, nullptr);
        acx_mgpu* mg = mr->mg;
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        const uint32_t W = mg->W;
        mr->witness_resident = false;                   // the resident witness is overwritten
        mr->h_valid = false;
        // second witness buffer per shard; released on EVERY exit (after the streams have drained)
        struct AltBuffers {
            acx_mgpu* mg;
            std::vector<uint4*> p;
            ~AltBuffers() {
                for (uint32_t s = 0; s < p.size(); ++s) {
                    if (!p[s]) continue;
                    (void)hipSetDevice(mg->sh[s].device);
                    (void)hipStreamSynchronize(mg->sh[s].ctx->stream);
                    (void)hipFree(p[s]);
                }
            }
        } altb{mg, std::vector<uint4*>(W, nullptr)};
        std::vector<uint4*>& alt = altb.p;
        for (uint32_t s = 0; s < W; ++s) {
            HIP_TRY(hipSetDevice(mg->sh[s].device));
            if (hipMalloc((void**)&alt[s], mr->m * 32) != hipSuccess) return fail(ACX_ERR_OOM, "device allocation failed");
        }
        for (auto& S : mg->sh) {                        // canonicity flag of the whole call
            HIP_TRY(hipSetDevice(S.device));
            HIP_TRY(hipMemsetAsync(S.d_res + 2, 0, 4, S.ctx->stream));
        }
        // This call's OWN result slots (section 2 of the ring; their reduction in section 3): slots that
        // acx_mgpu_r1cs_verify_enqueue has filled and acx_mgpu_r1cs_verdicts has not yet collected are left alone.
        const uint32_t kMany = 2 * 2 * kMgRing;         // word offset of the section
        std::vector<unsigned long long> ring_init(2 * kMgRing);
        for (uint32_t i = 0; i < kMgRing; ++i) { ring_init[2 * i] = 0; ring_init[2 * i + 1] = ~0ull; }
        auto reset_slots = [&]() {                      // error path: a later call must not see counts of this one
            for (uint32_t s = 0; s < W; ++s) {
                (void)hipSetDevice(mg->sh[s].device);
                (void)hipStreamSynchronize(mg->sh[s].ctx->stream);
                (void)hipMemcpy(mr->part[s].ring + kMany, ring_init.data(), 16 * kMgRing, hipMemcpyHostToDevice);
            }
        };
        int rc = ACX_OK;
        for (uint64_t done = 0; done < count && rc == ACX_OK; done += kMgRing) {
            const uint32_t k = (uint32_t)std::min<uint64_t>(kMgRing, count - done);
            rc = mg_per_shard_threads(mg, [&](uint32_t s) -> int {
                MgShard& S = mg->sh[s];
                HIP_TRY(hipSetDevice(S.device));
                CtxLock lock(S.ctx->mu);
                const auto& P = mr->part[s];
                for (uint32_t i = 0; i < k; ++i) {
                    uint4* d_w = ((done + i) & 1) ? alt[s] : P.d_w;
                    // the stream is in order: the check of witness i-2 (same buffer) precedes this copy
                    HIP_TRY(hipMemcpyAsync(d_w, witnesses + (done + i) * mr->m, mr->m * 32, hipMemcpyHostToDevice, S.ctx->stream));
                    ACX_TRY(launch_convert(S.ctx, true, d_w, d_w, mr->m, (uint32_t*)(S.d_res + 2)));
                    ACX_TRY(launch_residual(P.slab, d_w, P.row0, P.ring + kMany + 2 * i, nullptr, nullptr, 0));
                }
                return ACX_OK;
            });
            if (rc != ACX_OK) break;
            // ONE collective for the k verdicts (the body of acx_mgpu_r1cs_verdicts, slots 0 .. k-1)
            std::vector<unsigned long long> host(2 * k, 0), init(2 * k);
            for (uint32_t i = 0; i < k; ++i) { init[2 * i] = 0; init[2 * i + 1] = ~0ull; }
            std::vector<uint64_t> bad(k, 0);
            if (mg->rccl) {
                ncclResult_t r = mg->api->GroupStart();
                for (uint32_t s = 0; s < W && r == ncclSuccess; ++s)
                    r = mg->api->AllReduce(mr->part[s].ring + kMany, mr->part[s].ring + kMany + 2 * kMgRing, 2 * k, ncclUint64, ncclSum, mg->sh[s].comm, mg->sh[s].ctx->stream);
                const ncclResult_t r2 = mg->api->GroupEnd();
                if (r != ncclSuccess || r2 != ncclSuccess) { rc = fail(ACX_ERR_HIP, std::string("ncclAllReduce: ") + mg->api->GetErrorString(r != ncclSuccess ? r : r2)); break; }
                for (uint32_t s = 0; s < W; ++s) {
                    (void)hipSetDevice(mg->sh[s].device);
                    if (s == 0) (void)hipMemcpyAsync(host.data(), mr->part[0].ring + kMany + 2 * kMgRing, 16 * k, hipMemcpyDeviceToHost, mg->sh[0].ctx->stream);
                    (void)hipMemcpyAsync(mr->part[s].ring + kMany, init.data(), 16 * k, hipMemcpyHostToDevice, mg->sh[s].ctx->stream);
                }
                for (uint32_t s = 0; s < W; ++s) { (void)hipSetDevice(mg->sh[s].device); if (hipStreamSynchronize(mg->sh[s].ctx->stream) != hipSuccess) rc = fail(ACX_ERR_HIP, "stream synchronisation failed"); }
                for (uint32_t i = 0; i < k; ++i) bad[i] = host[2 * i];
            } else {
                std::vector<std::vector<unsigned long long>> per(W, std::vector<unsigned long long>(2 * k));
                for (uint32_t s = 0; s < W; ++s) {
                    (void)hipSetDevice(mg->sh[s].device);
                    (void)hipMemcpyAsync(per[s].data(), mr->part[s].ring + kMany, 16 * k, hipMemcpyDeviceToHost, mg->sh[s].ctx->stream);
                    (void)hipMemcpyAsync(mr->part[s].ring + kMany, init.data(), 16 * k, hipMemcpyHostToDevice, mg->sh[s].ctx->stream);
                }
                for (uint32_t s = 0; s < W; ++s) {
                    (void)hipSetDevice(mg->sh[s].device);
                    if (hipStreamSynchronize(mg->sh[s].ctx->stream) != hipSuccess) rc = fail(ACX_ERR_HIP, "stream synchronisation failed");
                    for (uint32_t i = 0; i < k; ++i) bad[i] += per[s][2 * i];
                }
            }
            for (uint32_t i = 0; i < k && rc == ACX_OK; ++i) {
                ok[done + i] = bad[i] == 0;
                if (n_bad) n_bad[done + i] = bad[i];
            }
        }
        if (rc == ACX_OK) {
            uint32_t flag = 0;
            (void)hipSetDevice(mg->sh[0].device);
            if (hipMemcpy(&flag, mg->sh[0].d_res + 2, 4, hipMemcpyDeviceToHost) != hipSuccess) rc = fail(ACX_ERR_HIP, "flag fetch failed");
            else if (flag) rc = fail(ACX_ERR_NONCANONICAL, "element >= p");
        }
        if (rc != ACX_OK) reset_slots();
        return rc;
    });
}

int acx_mgpu_qap_h_resident(acx_mgpu_r1cs* mr, const acx_fr* delta, int* ok) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        H256 dl[3];
        if (delta) for (int k = 0; k < 3; ++k) ACX_TRY(read_h256(&delta[k], mr->mg->sh[0].ctx->hf, dl[k]));
        std::lock_guard<std::mutex> g(mr->mg->mu);
        if (!mr->witness_resident) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        DevGuard dg;
        bool good = false;
        ACX_TRY(mg_qap_h_resident(mr, delta ? dl : nullptr, &good));
        *ok = good;
        return ACX_OK;
    });
}

// h of the resident witness from the devices' COLS blocks into natural order (caller holds mg->mu)
static int mg_qap_h_fetch_locked(acx_mgpu_r1cs* mr, acx_fr* out_h, uint64_t* h_len) {
    if (!mr->h_valid) return fail(ACX_ERR_UNSUPPORTED, "no h(x) on the devices (acx_mgpu_qap_h_resident)");
    acx_mgpu* mg = mr->mg;
    const uint64_t N = 1ull << mr->log_n, L = N / mg->W, R = 1ull << mr->log_r, C = N / R;
    ACX_TRY(mg_ensure_io(mg, L));
    std::vector<uint4*> hp(mg->W);
    for (uint32_t s = 0; s < mg->W; ++s) hp[s] = mr->part[s].vec + 2 * 7 * L;
    ACX_TRY(mg_fetch_natural(mg, hp.data(), R, C / mg->W, C, out_h));                         // COLS ownership
    write_h256(&out_h[N], mg->sh[0].ctx->hf, mr->h_top);
    uint64_t len = N + 1;
    static const uint8_t zero32[32] = {0};
    while (len > 0 && std::memcmp(out_h[len - 1].b, zero32, 32) == 0) --len;
    *h_len = len;
    return ACX_OK;
}

int acx_mgpu_qap_h_fetch(acx_mgpu_r1cs* mr, acx_fr* out_h, uint64_t* h_len) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !out_h || !h_len) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "no h(x) on the devices (acx_mgpu_qap_h_resident)");
        std::lock_guard<std::mutex> g(mr->mg->mu);
        DevGuard dg;
        return mg_qap_h_fetch_locked(mr, out_h, h_len);
    });
}

int acx_mgpu_qap_h(acx_mgpu_r1cs* mr, const acx_fr* witness, const acx_fr* delta, acx_fr* out_h, uint64_t* h_len, int* ok) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mr || !witness || !out_h || !h_len || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return acx_qap_h(mr->whole, witness, delta, out_h, h_len, ok);
        if (!mr->has_cyclic && !mr->verify_only) {
            // a transform size the distributed four-step form does not cover (N above 2^24, or fewer than 2 W points per
            // digit): the answer still comes, from ONE device on its copy of the whole system (the copies of
            // acx_mgpu_qap_columns) -- the handle is total over everything the single-GPU call accepts
            acx_r1cs* full = nullptr;
            {
                std::lock_guard<std::mutex> g(mr->mg->mu);
                DevGuard dg;
                ACX_TRY(mg_ensure_replicas(mr, false));
                full = mr->part[0].full;
            }
            return acx_qap_h(full, witness, delta, out_h, h_len, ok);
        }
        H256 dl[3];
        if (delta) for (int k = 0; k < 3; ++k) ACX_TRY(read_h256(&delta[k], mr->mg->sh[0].ctx->hf, dl[k]));
        // ONE critical section from the upload to the fetch: another thread's call on this handle cannot overwrite the
        // devices' vectors between the pipeline and the read-back
        std::lock_guard<std::mutex> g(mr->mg->mu);
        DevGuard dg;
        ACX_TRY(mg_upload_witness(mr, witness));
        bool good = false;
        const int rc = mg_qap_h_resident(mr, delta ? dl : nullptr, &good);
        if (rc != ACX_OK) { if (rc == ACX_ERR_NONCANONICAL) mr->witness_resident = false; return rc; }
        *ok = good;
        return mg_qap_h_fetch_locked(mr, out_h, h_len);
    });
}

int acx_mgpu_qap_columns(acx_mgpu_r1cs* mr, int matrix, uint64_t wire_begin, uint64_t wire_count, acx_fr* out, uint64_t* out_len) {
    ACX_RANGE();
    if (!mr || matrix < 0 || matrix > 2 || !out) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    if (wire_begin > mr->m || wire_count > mr->m - wire_begin) return fail(ACX_ERR_INVALID_ARG, "wire range exceeds m");
    if (wire_count == 0) return ACX_OK;
    if (!mr->sharded) return acx_qap_columns(mr->whole, matrix, wire_begin, wire_count, out, out_len);
    return guarded([&]() -> int {
        acx_mgpu* mg = mr->mg;
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        // one shard: its slab IS the whole system, and the single-GPU call builds the column view from it on the device
        if (mg->W == 1) return acx_qap_columns(mr->part[0].slab, matrix, wire_begin, wire_count, out, out_len);
        ACX_TRY(mg_ensure_col_slices(mr));
        // every shard interpolates the blocks of the range it owns, straight into the caller's buffers: no exchange at all.
        // Device-side batches are bounded so that all shards together stage at most 2 x 256 MiB of coefficients (at least one
        // column each), and the staging is released when the call returns (W contexts may share one device).
        const uint64_t N = 1ull << mr->log_n, W = mg->W, B = kMgColBlock, wire_end = wire_begin + wire_count;
        const uint64_t batch = std::max<uint64_t>(N * 32, (256ull << 20) / W);
        const int rc = mg_per_shard_threads(mg, [&](uint32_t s) -> int {
            for (uint64_t j = wire_begin / B; j * B < wire_end; ++j) {
                if (j % W != s) continue;
                const uint64_t lo = std::max(wire_begin, j * B), hi = std::min(wire_end, (j + 1) * B);
                ACX_TRY(qap_columns_host(mr->part[s].cols, matrix, (j / W) * B + (lo - j * B), hi - lo, out + (lo - wire_begin) * N,
                                         out_len ? out_len + (lo - wire_begin) : nullptr, batch));
            }
            return ACX_OK;
        });
        for (auto& S : mg->sh) ctx_trim_scratch(S.ctx);
        return rc;
    });
}

int acx_mgpu_ntt(acx_mgpu* mg, uint32_t log_n, int inverse, const acx_fr* shift, const acx_fr* in, acx_fr* out) {
    ACX_RANGE();
    return guarded([&]() -> int {
        if (!mg || !in || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mg_can_distribute(mg->W, log_n)) return acx_ntt(mg->sh[0].ctx, log_n, 1, inverse, shift, in, out);
        const HostField& hf = mg->sh[0].ctx->hf;
        if ((int)log_n > hf.two_adicity()) return fail(ACX_ERR_TOO_LARGE, "log_n exceeds the field's two-adicity");
        H256 sh;
        if (shift) {
            ACX_TRY(read_h256(shift, hf, sh));
            if (sh.is_zero()) return fail(ACX_ERR_INVALID_ARG, "coset shift must be nonzero");
        }
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        const uint32_t W = mg->W, log_r = log_n / 2;
        const uint64_t N = 1ull << log_n, L = N / W, R = 1ull << log_r, C = N / R;
        ACX_TRY(mg_ensure_slots(mg, L));
        ACX_TRY(mg_ensure_io(mg, L));
        // blocks: input in slot 1's send buffer, output in slot 1's recv buffer (slot 0 carries the transform)
        std::vector<uint4*> src(W), dst(W);
        for (uint32_t s = 0; s < W; ++s) { src[s] = mg->sh[s].slot[1].send; dst[s] = mg->sh[s].slot[1].recv; }
        // forward: COLS -> ROWS; inverse: ROWS -> COLS
        if (!inverse) ACX_TRY(mg_push_natural(mg, in, R, C / W, C, src.data())); else ACX_TRY(mg_push_natural(mg, in, C, R / W, R, src.data()));
        ACX_TRY(mg_check_canonical(mg));
        ACX_TRY(mg_per_shard_threads(mg, [&](uint32_t s) -> int {
            HIP_TRY(hipSetDevice(mg->sh[s].device));
            MgNtt nt(mg, log_n, log_r);
            ACX_TRY(nt.begin(s, 0, src[s], inverse, shift ? &sh : nullptr));
            return nt.finish(s, 0, dst[s], inverse, shift ? &sh : nullptr);
        }));
        if (!inverse) return mg_fetch_natural(mg, dst.data(), C, R / W, R, out);
        return mg_fetch_natural(mg, dst.data(), R, C / W, C, out);
    });
}

}  // extern "C"
