// abi_common.inc.h -- what every translation unit that defines entry points of include/acx.h shares: the thread-local
// error text, the exception barrier, the canonical <-> Montgomery edge of a field element, the host-side circuit handle.
// Pure host code (no HIP): included by engine.hip (libacx.so) and by host_only.cpp (the sanitizer build of the host
// marshalling code, tests/test_host_sanitized.py).
#pragma once
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>

#include "../../include/acx.h"
#include "circuit_host.h"
#include "host_field.h"

using namespace acx;

// ------------------------------------------------------------------------------------ errors
static thread_local std::string g_last_error;

static int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

// ACX_TRACE_LOAD=1: wall-clock of the phases of acx_r1cs_load / acx_circuit_to_r1cs on stderr (development aid)
struct PhaseTimer {
    bool on = std::getenv("ACX_TRACE_LOAD") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void mark(const char* what) {
        if (!on) return;
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[acx load] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

// Nothing may propagate through the C ABI: host allocations sized by caller data can throw.
template <class Fn>
static int guarded(Fn&& fn) {
    try {
        return fn();
    } catch (const std::bad_alloc&) {
        return fail(ACX_ERR_OOM, "host allocation failed");
    } catch (const std::exception& e) {
        return fail(ACX_ERR_INVALID_ARG, std::string("unexpected exception: ") + e.what());
    } catch (...) {
        return fail(ACX_ERR_INVALID_ARG, "unexpected exception");
    }
}


#define ACX_TRY(expr)            \
    do {                         \
        int rc_ = (expr);        \
        if (rc_ != ACX_OK) return rc_; \
    } while (0)


static int read_h256(const acx_fr* f, const HostField& hf, H256& mont) {
    H256 c;
    std::memcpy(c.l, f->b, 32);
    if (!hf.is_canonical(c)) return fail(ACX_ERR_NONCANONICAL, "element >= p");
    mont = hf.to_mont(c);
    return ACX_OK;
}

static void write_h256(acx_fr* f, const HostField& hf, const H256& mont) {
    const H256 c = hf.from_mont(mont);
    std::memcpy(f->b, c.l, 32);
}

struct acx_circuit {
    int field = 0;
    HostCircuit hc;
    HostCsr rows[3];     // gateToGenQAP rows in gate order, built once
    // A system built from this circuit derives its device evaluation plan (acx_r1cs_eval) lazily, on first use, and
    // holds a reference until then: acx_circuit_destroy releases the rows at once and the gate list with the last reference.
    mutable std::atomic<int> refs{1};
};
static void circuit_release(const acx_circuit* c) {
    if (c && c->refs.fetch_sub(1) == 1) delete c;
}

