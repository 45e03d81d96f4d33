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

constexpr int kMgSlots = 3;          // transforms in flight (the three vectors of h(x))
constexpr uint32_t kMgRing = 16;     // result slots of acx_mgpu_r1cs_verify_enqueue

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
        acx_r1cs* full = nullptr;                   // the WHOLE system, for this shard's wires of acx_mgpu_qap_columns (built on its first call)
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

// ---- one distributed transform = begin (local step 0 + the START of the exchange) and finish (wait + local step 1) ----
struct MgNtt {
    acx_mgpu* mg;
    uint32_t log_n, log_r;
    uint64_t L, chunk;                              // elements per shard; per (shard, peer) block
    MgNtt(acx_mgpu* m, uint32_t ln, uint32_t lr) : mg(m), log_n(ln), log_r(lr) {
        L = (1ull << ln) / m->W;
        chunk = L / m->W;
    }

    int exchange(int k) {
        const uint32_t W = mg->W;
        if (mg->rccl) {
            for (auto& s : mg->sh) {
                HIP_TRY(hipSetDevice(s.device));
                MgSlot& sl = s.slot[k];
                HIP_TRY(hipStreamWaitEvent(s.xstream, sl.sent, 0));
                if (sl.used_valid) HIP_TRY(hipStreamWaitEvent(s.xstream, sl.used, 0));       // previous reader of recv
            }
            NCCL_TRY(mg, mg->api->GroupStart());
            for (auto& s : mg->sh) {
                MgSlot& sl = s.slot[k];
                const ncclResult_t r = mg->api->AllToAll(sl.send, sl.recv, chunk * 4, ncclUint64, s.comm, s.xstream);
                if (r != ncclSuccess) { (void)mg->api->GroupEnd(); return fail(ACX_ERR_HIP, std::string("ncclAllToAll: ") + mg->api->GetErrorString(r)); }
            }
            NCCL_TRY(mg, mg->api->GroupEnd());
            for (auto& s : mg->sh) {
                HIP_TRY(hipSetDevice(s.device));
                HIP_TRY(hipEventRecord(s.slot[k].got, s.xstream));
                s.slot[k].got_valid = true;
            }
            return ACX_OK;
        }
        // peer copies: shard t PULLS block t of every shard's send buffer
        for (uint32_t t = 0; t < W; ++t) {
            MgShard& dst = mg->sh[t];
            HIP_TRY(hipSetDevice(dst.device));
            MgSlot& dl = dst.slot[k];
            if (dl.used_valid) HIP_TRY(hipStreamWaitEvent(dst.xstream, dl.used, 0));
            for (uint32_t s = 0; s < W; ++s) {
                MgShard& src = mg->sh[s];
                HIP_TRY(hipStreamWaitEvent(dst.xstream, src.slot[k].sent, 0));
                uint4* to = dl.recv + 2 * (uint64_t)s * chunk;
                const uint4* from = src.slot[k].send + 2 * (uint64_t)t * chunk;
                if (src.device == dst.device) HIP_TRY(hipMemcpyAsync(to, from, chunk * 32, hipMemcpyDeviceToDevice, dst.xstream));
                else HIP_TRY(hipMemcpyPeerAsync(to, dst.device, from, src.device, chunk * 32, dst.xstream));
            }
            HIP_TRY(hipEventRecord(dl.got, dst.xstream));
            dl.got_valid = true;
        }
        return ACX_OK;
    }

    // in[s]: L dev elements per shard (COLS for a forward, ROWS for an inverse transform)
    // rows_transposed: the input of an inverse transform is in ascending row order [k2][kl] (the residual kernel's dots)
    // mul: the transform of the pointwise product in[s][i] * mul[s][i] (same layout)
    int begin(int k, uint4* const* in, int inverse, const H256* shift, bool rows_transposed = false, uint4* const* mul = nullptr) {
        const uint32_t W = mg->W;
        for (uint32_t s = 0; s < W; ++s) {
            MgShard& S = mg->sh[s];
            HIP_TRY(hipSetDevice(S.device));
            CtxLock lock(S.ctx->mu);
            if (!mg->rccl)                                          // peers still pulling the previous contents of send
                for (uint32_t t = 0; t < W; ++t)
                    if (mg->sh[t].slot[k].got_valid) HIP_TRY(hipStreamWaitEvent(S.ctx->stream, mg->sh[t].slot[k].got, 0));
            ACX_TRY(ntt_dist_step_locked(S.ctx, log_n, log_r, W, s, inverse, 0, shift, in[s], S.slot[k].send, rows_transposed,
                                         mul ? mul[s] : nullptr));
            HIP_TRY(hipEventRecord(S.slot[k].sent, S.ctx->stream));
        }
        return exchange(k);
    }

    // add: out[s][k] = X[k] + add[s][k] (same layout as out)
    int finish(int k, uint4* const* out, int inverse, const H256* shift, uint4* const* add = nullptr) {
        const uint32_t W = mg->W;
        for (uint32_t s = 0; s < W; ++s) {
            MgShard& S = mg->sh[s];
            HIP_TRY(hipSetDevice(S.device));
            CtxLock lock(S.ctx->mu);
            HIP_TRY(hipStreamWaitEvent(S.ctx->stream, S.slot[k].got, 0));
            ACX_TRY(ntt_dist_step_locked(S.ctx, log_n, log_r, W, s, inverse, 1, shift, S.slot[k].recv, out[s], false, nullptr,
                                         add ? add[s] : nullptr));
            HIP_TRY(hipEventRecord(S.slot[k].used, S.ctx->stream));
            S.slot[k].used_valid = true;
        }
        return ACX_OK;
    }
};

// run fn(shard) on one host thread per shard (host-to-device copies of pageable memory block their caller: one thread per
// PCIe link); the first failure and its message are carried back to the calling thread
template <class Fn>
int mg_per_shard_threads(acx_mgpu* mg, Fn&& fn) {
    const uint32_t W = mg->W;
    std::vector<int> rc(W, ACX_OK);
    std::vector<std::string> msg(W);
    auto body = [&](uint32_t s) {
        try {
            rc[s] = fn(s);
        } catch (const std::bad_alloc&) {
            rc[s] = fail(ACX_ERR_OOM, "host allocation failed");
        } catch (...) {
            rc[s] = fail(ACX_ERR_INVALID_ARG, "unexpected exception");
        }
        if (rc[s] != ACX_OK) msg[s] = g_last_error;
    };
    if (W == 1) {
        body(0);
    } else {
        std::vector<std::thread> th;
        uint32_t started = 0;
        try {
            th.reserve(W);
            for (; started < W; ++started) th.emplace_back(body, started);
        } catch (...) {
        }
        for (uint32_t s = started; s < W; ++s) body(s);
        for (auto& t : th) t.join();
    }
    for (uint32_t s = 0; s < W; ++s)
        if (rc[s] != ACX_OK) return fail(rc[s], msg[s]);
    return ACX_OK;
}

// replicate the witness: upload + conversion on every shard; the canonicity flag lands in the shard's CallSlot
int mg_upload_witness(acx_mgpu_r1cs* mr, const acx_fr* witness) {
    acx_mgpu* mg = mr->mg;
    mr->witness_resident = false;
    mr->h_valid = false;
    ACX_TRY(mg_per_shard_threads(mg, [&](uint32_t s) -> int {
        MgShard& S = mg->sh[s];
        HIP_TRY(hipSetDevice(S.device));
        CtxLock lock(S.ctx->mu);
        static const CallSlot init{0ull, ~0ull, 0u, {0u, 0u, 0u}};
        HIP_TRY(hipMemcpyAsync(S.d_res, &init, sizeof(init), hipMemcpyHostToDevice, S.ctx->stream));
        uint4* d_w = mr->part[s].d_w;
        HIP_TRY(hipMemcpyAsync(d_w, witness, mr->m * 32, hipMemcpyHostToDevice, S.ctx->stream));
        return launch_convert(S.ctx, true, d_w, d_w, mr->m, (uint32_t*)(S.d_res + 2));
    }));
    mr->witness_resident = true;
    return ACX_OK;
}

// residual launch on every shard (+ dots when the h(x) pipeline follows) and the verdict.
// Two halves, so that h(x) can issue its whole pipeline between them and the host waits once, at the end.
// mg_residual_enqueue: the residual launch on every shard (+ dots when the h(x) pipeline follows) and, with RCCL, THE verdict
// collective behind it -- everything asynchronous.  mg_residual_fetch: the verdict (one wait).
int mg_residual_enqueue(acx_mgpu_r1cs* mr, bool with_dots, bool scaled_dots = false) {
    acx_mgpu* mg = mr->mg;
    const uint32_t W = mg->W;
    const uint64_t L = (1ull << mr->log_n) / W, rw = (1ull << mr->log_r) / W;
    for (uint32_t s = 0; s < W; ++s) {
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
    }
    if (mg->rccl) {
        // THE verdict collective: sum of the violated-row counts, into word 4 of every shard's slot
        NCCL_TRY(mg, mg->api->GroupStart());
        for (auto& S : mg->sh) {
            const ncclResult_t r = mg->api->AllReduce(S.d_res, S.d_res + 4, 1, ncclUint64, ncclSum, S.comm, S.ctx->stream);
            if (r != ncclSuccess) { (void)mg->api->GroupEnd(); return fail(ACX_ERR_HIP, std::string("ncclAllReduce: ") + mg->api->GetErrorString(r)); }
        }
        NCCL_TRY(mg, mg->api->GroupEnd());
    }
    return ACX_OK;
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

int mg_residual(acx_mgpu_r1cs* mr, bool with_dots, bool want_first, uint64_t* n_bad, uint64_t* first_bad, bool* noncanonical) {
    ACX_TRY(mg_residual_enqueue(mr, with_dots));
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

// Per-wire polynomials (`createPolynomialsFFT`, src/QAP.hs:512-525) shard by WIRE with no communication: a
// column's interpolation needs every row of its matrix, so each shard takes a copy of the whole system.  Only callers
// of acx_mgpu_qap_columns pay for that, on their first call: the slabs are read back from the devices (canonical CSR,
// acx_r1cs_export), joined on the host and loaded on every shard by one thread each.
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

// verificationWitnessZk over the shards on the resident witness; h stays on the devices in COLS ownership
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
    const bool zk = dl && !(dl[0].is_zero() && dl[1].is_zero() && dl[2].is_zero());
    // without the zero-knowledge terms 1/z and -1/z ride on the stored dots, the last transform takes (L/z) * R as its first
    // step loads the points and adds -O/z behind its closing step (qap_h_dev_locked's fused form, sharded)
    const bool fusedh = !zk && mr->part[0].hscale != nullptr;
    ACX_TRY(mg_residual_enqueue(mr, true, fusedh)); // the verdict is fetched after the whole pipeline has been issued: one wait
    const H256 g = hf.generator();
    MgNtt nt(mg, mr->log_n, mr->log_r);
    auto ptrs = [&](uint64_t off) { std::vector<uint4*> v(W); for (uint32_t s = 0; s < W; ++s) v[s] = mr->part[s].vec + 2 * off; return v; };
    // vec: dots k at k L (ROWS), coefficients k at (3 + k) L (COLS), pointwise product at 6 L (ROWS), h at 7 L (COLS)
    // Software pipeline over the three vectors (qap_h_dev_locked's sequence, sharded): vector k's exchange runs on the
    // exchange streams under vector k+1's local step, and a vector's coset transform starts as soon as its inverse one is
    // complete -- of the six all-to-alls only the last has no local work to hide behind.
    for (int k = 0; k < 3; ++k) ACX_TRY(nt.begin(k, ptrs((uint64_t)k * L).data(), 1, nullptr, true));     // dots: ascending row order
    for (int k = 0; k < 3; ++k) {
        ACX_TRY(nt.finish(k, ptrs((3 + (uint64_t)k) * L).data(), 1, nullptr));
        if (k < 2) ACX_TRY(nt.begin(k, ptrs((3 + (uint64_t)k) * L).data(), 0, &g));
    }
    for (int k = 0; k < 2; ++k) ACX_TRY(nt.finish(k, ptrs((uint64_t)k * L).data(), 0, &g));
    const H256 zinv = hf.inv(hf.sub(hf.pow_u64(g, N), hf.one()));
    const H256 mzinv = hf.sub(hf.zero(), zinv);
    if (fusedh) {
        ACX_TRY(nt.begin(0, ptrs(0).data(), 1, &g, false, ptrs(L).data()));
        ACX_TRY(nt.finish(0, ptrs(7 * L).data(), 1, &g, ptrs(5 * L).data()));
    } else {
        for (uint32_t s = 0; s < W; ++s) {
            MgShard& S = mg->sh[s];
            HIP_TRY(hipSetDevice(S.device));
            CtxLock lock(S.ctx->mu);
            uint4* v = mr->part[s].vec;
            DISPATCH_FIELD(S.ctx, hipLaunchKernelGGL((k_pointwise_h<F>), dim3(grid_for(S.ctx, L)), dim3(kBlock), 0, S.ctx->stream, (const uint4*)v,
                                                     (const uint4*)(v + 2 * L), (const uint4*)nullptr, v + 2 * 6 * L, L, dev_arg(hf, zinv), 0u));
            HIP_TRY(hipGetLastError());
        }
        ACX_TRY(nt.begin(0, ptrs(6 * L).data(), 1, &g));
        ACX_TRY(nt.finish(0, ptrs(7 * L).data(), 1, &g));
    }
    for (uint32_t s = 0; s < W && !fusedh; ++s) {
        MgShard& S = mg->sh[s];
        HIP_TRY(hipSetDevice(S.device));
        CtxLock lock(S.ctx->mu);
        uint4* v = mr->part[s].vec;
        uint4 *h = v + 2 * 7 * L, *L0 = v + 2 * 3 * L, *R0 = v + 2 * 4 * L, *O0 = v + 2 * 5 * L;
        if (zk) {
            // (L0+d1 T)(R0+d2 T) - (O0+d3 T) = T (h0 + d1 R0 + d2 L0 + d1 d2 T - d3), T = x^N - 1 (src/QAP.hs:315-323): the
            // elementwise part is layout agnostic (h, L0, R0, O0 share the COLS ownership); coefficient 0 lives on shard 0
            // at local index 0 and coefficient N (= d1 d2) is appended by the fetch
            const H256 d12 = hf.mul(dl[0], dl[1]);
            DISPATCH_FIELD(S.ctx, {
                hipLaunchKernelGGL((k_axpy3<F>), dim3(grid_for(S.ctx, L)), dim3(kBlock), 0, S.ctx->stream, h, (const uint4*)R0, (const uint4*)L0,
                                   (const uint4*)O0, L, dev_arg(hf, dl[0]), dev_arg(hf, dl[1]), dev_arg(hf, mzinv));
                if (s == 0) hipLaunchKernelGGL((k_h_fix<F>), dim3(1), dim3(64), 0, S.ctx->stream, h, ~(u64)0, dev_arg(hf, hf.add(d12, dl[2])),
                                               dev_arg(hf, hf.zero()));
            });
        } else {
            DISPATCH_FIELD(S.ctx, hipLaunchKernelGGL((k_axpy3<F>), dim3(grid_for(S.ctx, L)), dim3(kBlock), 0, S.ctx->stream, h, (const uint4*)nullptr,
                                                     (const uint4*)nullptr, (const uint4*)O0, L, dev_arg(hf, mzinv), dev_arg(hf, mzinv), dev_arg(hf, mzinv)));
        }
        HIP_TRY(hipGetLastError());
    }
    uint64_t n_bad = 0, first = 0;
    bool noncanon = false;
    ACX_TRY(mg_residual_fetch(mr, false, &n_bad, &first, &noncanon));
    if (noncanon) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    mr->h_top = zk ? hf.mul(dl[0], dl[1]) : hf.zero();
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
    if (!mg || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    const acx_csr* mats[3] = {A, B, C};
    std::lock_guard<std::mutex> g(mg->mu);
    DevGuard dg;
    return guarded([&]() -> int { return mg_load(mg, n, m, mats, flags, out); });
}

int acx_mgpu_circuit_to_r1cs(acx_mgpu* mg, const acx_circuit* c, const acx_fr* roots, uint64_t n_roots, uint32_t flags,
                             acx_mgpu_r1cs** out) {
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
    return guarded([&]() -> int {
        if (!mr || !witness || !ok) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return acx_r1cs_verify(mr->whole, witness, ok, n_bad, first_bad);
        std::lock_guard<std::mutex> g(mr->mg->mu);
        DevGuard dg;
        ACX_TRY(mg_upload_witness(mr, witness));
        uint64_t bad = 0, first = ~0ull;
        bool noncanon = false;
        ACX_TRY(mg_residual(mr, false, first_bad != nullptr, &bad, &first, &noncanon));
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
    return guarded([&]() -> int {
        if (!mr || slot >= kMgRing) return fail(ACX_ERR_INVALID_ARG, "bad argument (slot < 16)");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        acx_mgpu* mg = mr->mg;
        std::lock_guard<std::mutex> g(mg->mu);
        if (!mr->witness_resident) return fail(ACX_ERR_UNSUPPORTED, "no resident witness (acx_mgpu_witness_upload) on a sharded system");
        DevGuard dg;
        for (uint32_t s = 0; s < mg->W; ++s) {
            MgShard& S = mg->sh[s];
            HIP_TRY(hipSetDevice(S.device));
            CtxLock lock(S.ctx->mu);
            const auto& P = mr->part[s];
            ACX_TRY(launch_residual(P.slab, P.d_w, P.row0, P.ring + 2 * slot, nullptr, nullptr, 0));
        }
        return ACX_OK;
    });
}

int acx_mgpu_r1cs_verdicts(acx_mgpu_r1cs* mr, uint32_t slot0, uint32_t count, uint64_t* n_bad) {
    return guarded([&]() -> int {
        if (!mr || !n_bad || count == 0 || slot0 + count > kMgRing) return fail(ACX_ERR_INVALID_ARG, "bad argument (slot0 + count <= 16)");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "system is held whole on shard 0");
        acx_mgpu* mg = mr->mg;
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        const uint32_t W = mg->W;
        std::vector<unsigned long long> host(2 * count), init(2 * count);
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
        std::vector<std::vector<unsigned long long>> per(W, std::vector<unsigned long long>(2 * count));
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
    return guarded([&]() -> int {
        if (!mr || !out_h || !h_len) return fail(ACX_ERR_INVALID_ARG, "null argument");
        if (!mr->sharded) return fail(ACX_ERR_UNSUPPORTED, "no h(x) on the devices (acx_mgpu_qap_h_resident)");
        std::lock_guard<std::mutex> g(mr->mg->mu);
        DevGuard dg;
        return mg_qap_h_fetch_locked(mr, out_h, h_len);
    });
}

int acx_mgpu_qap_h(acx_mgpu_r1cs* mr, const acx_fr* witness, const acx_fr* delta, acx_fr* out_h, uint64_t* h_len, int* ok) {
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
    if (!mr || matrix < 0 || matrix > 2 || !out) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    if (wire_begin > mr->m || wire_count > mr->m - wire_begin) return fail(ACX_ERR_INVALID_ARG, "wire range exceeds m");
    if (wire_count == 0) return ACX_OK;
    if (!mr->sharded) return acx_qap_columns(mr->whole, matrix, wire_begin, wire_count, out, out_len);
    return guarded([&]() -> int {
        acx_mgpu* mg = mr->mg;
        std::lock_guard<std::mutex> g(mg->mu);
        DevGuard dg;
        ACX_TRY(mg_ensure_replicas(mr));
        // contiguous wire ranges of equal size, one per shard, straight into the caller's buffers: no exchange at all
        const uint64_t N = 1ull << mr->log_n, W = mg->W;
        return mg_per_shard_threads(mg, [&](uint32_t s) -> int {
            const uint64_t w0 = wire_count * s / W, w1 = wire_count * (s + 1) / W;
            if (w0 == w1) return ACX_OK;
            return acx_qap_columns(mr->part[s].full, matrix, wire_begin + w0, w1 - w0, out + w0 * N, out_len ? out_len + w0 : nullptr);
        });
    });
}

int acx_mgpu_ntt(acx_mgpu* mg, uint32_t log_n, int inverse, const acx_fr* shift, const acx_fr* in, acx_fr* out) {
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
        MgNtt nt(mg, log_n, log_r);
        ACX_TRY(nt.begin(0, src.data(), inverse, shift ? &sh : nullptr));
        ACX_TRY(nt.finish(0, dst.data(), inverse, shift ? &sh : nullptr));
        if (!inverse) return mg_fetch_natural(mg, dst.data(), C, R / W, R, out);
        return mg_fetch_natural(mg, dst.data(), R, C / W, C, out);
    });
}

}  // extern "C"
