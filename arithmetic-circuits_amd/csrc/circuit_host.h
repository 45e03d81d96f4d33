// circuit_host.h -- host-side marshalling of the reference's circuit types into constraint rows.
//
// Restates, in C++ over flat arrays (all paths into /root/reference):
//   affineCircuitToAffineMap  src/Circuit/Affine.hs:90-105
//   evalAffineCircuit         src/Circuit/Affine.hs:73-86
//   evalGate/evalArithCircuit src/Circuit/Arithmetic.hs:106-145,221-235
//   validArithCircuit         src/Circuit/Arithmetic.hs:158-185
//   generateRoots (row count) src/Circuit/Arithmetic.hs:194-216
//   gateToGenQAP              src/QAP.hs:366-474   (row contents: SURVEY.md Appendix A.2)
//   qapSetToMap numbering     src/QAP.hs:605-620
#pragma once
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <array>
#include <atomic>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <new>
#include <thread>
#include <map>
#include <string>
#include <vector>

#include "../../include/acx.h"
#include "host_field.h"

namespace acx {

// Worker threads for the host-side loops over gates / scalars (row generation is the reference's
// arithCircuitToGenQAP, src/QAP.hs:530-539).  ACX_HOST_THREADS overrides the count.
// CPUs this process may really use: the hardware threads, cut by a cgroup CPU quota when there is one (cpu.max = "quota
// period"; the GPU boxes show 256 threads under a quota of 16: more busy threads than that are only throttled)
inline unsigned usable_cpus() {
    static const unsigned cached = [] {
        unsigned n = std::max(1u, std::thread::hardware_concurrency());
        if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
            char q[32] = {0};
            unsigned long long period = 0;
            if (std::fscanf(f, "%31s %llu", q, &period) == 2 && period > 0 && std::strcmp(q, "max") != 0) {
                const unsigned long long quota = std::strtoull(q, nullptr, 10);
                if (quota > 0) n = (unsigned)std::min<unsigned long long>(n, std::max<unsigned long long>(1, (quota + period / 2) / period));
            }
            std::fclose(f);
        }
        return n;
    }();
    return cached;
}
inline unsigned host_threads(uint64_t items, uint64_t grain) {
    unsigned t = std::min<unsigned>({usable_cpus(), 64u, (unsigned)(items / grain + 1)});
    if (const char* e = std::getenv("ACX_HOST_THREADS")) t = (unsigned)std::min(256, std::max(1, std::atoi(e)));
    return t;
}
// body(t, begin, end) over a partition of [0, n) into T contiguous ranges.  Nothing may escape a worker thread or leave a
// joinable std::thread behind (either is std::terminate, which would cross the C ABI): a worker's exception is carried to
// the caller and rethrown after the join, and ranges whose thread could not be created (std::system_error at the thread
// limit) run on the calling thread.
template <class Body>
inline void parallel_ranges(uint64_t n, unsigned T, Body&& body) {
    if (T <= 1) { body(0u, (uint64_t)0, n); return; }
    std::exception_ptr err;
    std::mutex err_mu;
    auto run = [&](unsigned t) {
        try {
            body(t, n * t / T, n * (t + 1) / T);
        } catch (...) {
            std::lock_guard<std::mutex> g(err_mu);
            if (!err) err = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    unsigned started = 0;
    try {
        th.reserve(T);
        for (; started < T; ++started) th.emplace_back(run, started);
    } catch (...) {
        // fall through: the ranges [started, T) run inline below
    }
    for (unsigned t = started; t < T; ++t) run(t);
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
}

// std::vector whose resize() leaves trivially constructible elements uninitialised: the big arrays of a circuit (hundreds
// of MB at 2^20 gates) are filled by worker threads right after being sized, and value-initialising them first means ONE
// thread touching every page -- a quarter of acx_circuit_create at 8 threads (ACX_TRACE_LOAD).
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U, class... Args>
    void construct(U* p, Args&&... args) {
        if constexpr (sizeof...(Args) == 0) ::new ((void*)p) U;            // default-initialisation: nothing for trivial types
        else ::new ((void*)p) U(std::forward<Args>(args)...);
    }
};
template <class T> using RawVec = std::vector<T, NoInitAlloc<T>>;

// dst = src[0 .. n) copied (and first touched) by the worker threads
template <class T>
inline void parallel_copy(RawVec<T>& dst, const T* src, uint64_t n) {
    dst.resize(n);
    parallel_ranges(n, host_threads(n * sizeof(T), 4u << 20), [&](unsigned, uint64_t b, uint64_t e) {
        if (e > b) std::memcpy(dst.data() + b, src + b, (e - b) * sizeof(T));
    });
}

struct HostCsr {
    std::vector<uint32_t> rowptr{0};
    RawVec<uint32_t> col;
    RawVec<H256> val;  // canonical
    template <class Row>     // any range of (column, CANONICAL value) pairs in ascending column order
    void push_row(const HostField&, const Row& row) {
        for (const auto& kv : row) {
            if (kv.second.is_zero()) continue;  // explicit zeros of the reference are numerically void
            col.push_back((uint32_t)kv.first);
            val.push_back(kv.second);
        }
        rowptr.push_back((uint32_t)col.size());
    }
};

class HostCircuit {
public:
    HostField hf;
    uint64_t n_gates = 0;
    RawVec<uint8_t> kind;
    RawVec<uint64_t> tok_ofs;
    RawVec<uint8_t> tok_op;
    RawVec<uint32_t> tok_arg;
    RawVec<H256> scalars;  // canonical
    mutable RawVec<H256> scalars_mont;         // Montgomery copy for the host fold (eval), made on first use
    mutable std::once_flag scalars_mont_once;
    const H256* mont_scalars() const {
        std::call_once(scalars_mont_once, [&] {
            scalars_mont.resize(scalars.size());
            parallel_ranges(scalars.size(), host_threads(scalars.size(), 1 << 16), [&](unsigned, uint64_t b, uint64_t e) {
                for (uint64_t i = b; i < e; ++i) scalars_mont[i] = hf.to_mont(scalars[i]);
            });
        });
        return scalars_mont.data();
    }
    // canonical constants and the canonical product of two canonical values ((a b / R) R^2 / R)
    static H256 c_one() { return H256{{1, 0, 0, 0}}; }
    H256 cmul(const H256& a, const H256& b) const { return hf.to_mont(hf.mul(a, b)); }
    RawVec<acx_wire> aff_wires;
    RawVec<uint64_t> wire_ofs;
    RawVec<acx_wire> wires;
    uint64_t n_in = 0, n_mid = 0, n_out = 0;

    uint64_t m() const { return 1 + n_in + n_mid + n_out; }
    uint64_t flat(const acx_wire& w) const {
        switch (w.kind) {
            case ACX_WIRE_INPUT: return 1 + w.index;
            case ACX_WIRE_INTERMEDIATE: return 1 + n_in + w.index;
            default: return 1 + n_in + n_mid + w.index;
        }
    }
    uint64_t rows_of_gate(uint64_t g) const {
        switch (kind[g]) {
            case ACX_GATE_MUL: return 1;
            case ACX_GATE_EQUAL: return 2;
            default: return wire_ofs[g + 1] - wire_ofs[g];  // 1 + #outputs
        }
    }
    uint64_t n_rows() const {
        uint64_t n = 0;
        for (uint64_t g = 0; g < n_gates; ++g) n += rows_of_gate(g);
        return n;
    }

    // Copies + validates the marshalled list; returns ACX_OK or an error with msg set.
    int init(const acx_gate_list* gl, std::string& msg) {
        if (!gl) { msg = "null gate list"; return ACX_ERR_INVALID_ARG; }
        n_gates = gl->n_gates;
        if (n_gates && (!gl->kind || !gl->tok_ofs || !gl->wire_ofs)) { msg = "null gate arrays"; return ACX_ERR_INVALID_ARG; }
        if (n_gates == 0) {                       // the empty circuit: the caller may pass NULL arrays
            kind.clear(); tok_ofs = {0}; wire_ofs = {0};
        } else {
            parallel_copy(kind, gl->kind, n_gates);
            parallel_copy(tok_ofs, gl->tok_ofs, 2 * n_gates + 1);
            parallel_copy(wire_ofs, gl->wire_ofs, n_gates + 1);
        }
        const uint64_t n_tok = tok_ofs.back(), n_w = wire_ofs.back();
        if ((n_tok && (!gl->tok_op || !gl->tok_arg)) || (n_w && !gl->wires) || (gl->n_aff_wires && !gl->aff_wires) ||
            (gl->n_scalars && !gl->scalars)) { msg = "null array with a nonzero count"; return ACX_ERR_INVALID_ARG; }
        for (size_t i = 0; i + 1 < tok_ofs.size(); ++i)
            if (tok_ofs[i] > tok_ofs[i + 1]) { msg = "tok_ofs not monotone"; return ACX_ERR_BAD_CIRCUIT; }
        for (size_t i = 0; i + 1 < wire_ofs.size(); ++i)
            if (wire_ofs[i] > wire_ofs[i + 1]) { msg = "wire_ofs not monotone"; return ACX_ERR_BAD_CIRCUIT; }
        if (n_tok) { parallel_copy(tok_op, gl->tok_op, n_tok); parallel_copy(tok_arg, gl->tok_arg, n_tok); }
        if (n_w) parallel_copy(wires, gl->wires, n_w);
        if (gl->n_aff_wires) parallel_copy(aff_wires, gl->aff_wires, gl->n_aff_wires);
        // The scalars stay CANONICAL, as the caller gave them and as the rows leave this class: gateToGenQAP only adds
        // coefficients, and multiplies two of them where ScalarMul nodes nest (cmul: two Montgomery products there) -- a flat
        // side s * Var x costs no multiplication at all, where converting every scalar in and every row entry out cost two
        // (2/3 of acx_circuit_create on the reference's benchmark circuit).  The host fold wants Montgomery operands and makes
        // its own copy on first use (mont_scalars).
        scalars.resize(gl->n_scalars);
        {
            std::atomic<bool> noncanonical{false};
            parallel_ranges(gl->n_scalars, host_threads(gl->n_scalars, 1 << 16), [&](unsigned, uint64_t b, uint64_t e) {
                for (uint64_t i = b; i < e; ++i) {
                    H256 c;
                    std::memcpy(c.l, gl->scalars[i].b, 32);
                    if (!hf.is_canonical(c)) { noncanonical = true; return; }
                    scalars[i] = c;
                }
            });
            if (noncanonical) { msg = "scalar >= p"; return ACX_ERR_NONCANONICAL; }
        }
        auto bump = [&](const acx_wire& w) -> bool {
            if (w.kind > ACX_WIRE_OUTPUT || w.index >= 0x7fffffffu) return false;
            uint64_t& d = w.kind == ACX_WIRE_INPUT ? n_in : (w.kind == ACX_WIRE_INTERMEDIATE ? n_mid : n_out);
            d = std::max<uint64_t>(d, (uint64_t)w.index + 1);
            return true;
        };
        for (const auto& w : wires) if (!bump(w)) { msg = "bad wire"; return ACX_ERR_BAD_CIRCUIT; }
        for (const auto& w : aff_wires) if (!bump(w)) { msg = "bad wire"; return ACX_ERR_BAD_CIRCUIT; }
        const unsigned T = host_threads(n_gates, 1 << 14);
        std::vector<const char*> err(T, nullptr);            // first problem of every gate range; the lowest range reports
        parallel_ranges(n_gates, T, [&](unsigned t, uint64_t gb, uint64_t ge) {
            for (uint64_t g = gb; g < ge; ++g) {
                const uint64_t nw = wire_ofs[g + 1] - wire_ofs[g];
                const bool has_tok = tok_ofs[2 * g + 2] > tok_ofs[2 * g];
                if (kind[g] == ACX_GATE_MUL) {
                    if (nw != 1) { err[t] = "Mul gate needs exactly one wire"; return; }
                    for (int side = 0; side < 2; ++side) {
                        uint64_t pos = tok_ofs[2 * g + side];
                        if (!check_tree(pos, tok_ofs[2 * g + side + 1], gl) || pos != tok_ofs[2 * g + side + 1]) {
                            err[t] = "malformed affine token stream"; return;
                        }
                    }
                } else if (kind[g] == ACX_GATE_EQUAL) {
                    if (nw != 3 || has_tok) { err[t] = "Equal gate needs three wires"; return; }
                } else if (kind[g] == ACX_GATE_SPLIT) {
                    if (nw < 1 || has_tok) { err[t] = "Split gate needs an input wire"; return; }
                } else { err[t] = "unknown gate kind"; return; }
            }
        });
        for (const char* e : err) if (e) { msg = e; return ACX_ERR_BAD_CIRCUIT; }
        if (m() >= 0xffffffffull) { msg = "too many wires"; return ACX_ERR_TOO_LARGE; }
        return ACX_OK;
    }

    // One row of one matrix: (column, value) pairs, ascending and unique by column once finished.  A flat vector, not a
    // std::map: a gate's row holds a handful of entries and is built 3 * 2^20 times for a 2^20-gate circuit.
    using Entries = std::vector<std::pair<uint64_t, H256>>;
    static void set_entry(Entries& row, uint64_t col, const H256& v) {       // Map.insert: a later value for the same wire overwrites
        for (auto& kv : row) if (kv.first == col) { kv.second = v; return; }
        row.emplace_back(col, v);
    }
    static void sort_entries(Entries& row) {
        std::sort(row.begin(), row.end(), [](const std::pair<uint64_t, H256>& x, const std::pair<uint64_t, H256>& y) { return x.first < y.first; });
    }

    // affineCircuitToAffineMap of the sub-tree whose pre-order tokens are [begin, end) (src/Circuit/Affine.hs:90-105), in ONE
    // left-to-right pass with a stack of pending scale factors instead of a map per node: a node is reached with the product s
    // of the ScalarMul factors above it; Add hands s to both sub-trees, ScalarMul c hands s c to its sub-tree, ConstGate c adds
    // s c to the constant, Var x emits the term (x, s).  Scaling distributes over Map.unionWith (+) exactly in a field, so the
    // terms of one wire, summed at the end, are the reference's map entry -- including the wire whose coefficients cancel, which
    // stays with value 0.  No recursion and no allocation per node: a 10^5-term Add chain costs one stack of scales.
    // `vec` comes back ascending and unique by column; coefficients and `cst` canonical.
    void affine_map(uint64_t begin, uint64_t end, H256& cst, Entries& vec) const {
        static thread_local std::vector<H256> st;
        st.clear();
        const H256 one = c_one();
        st.push_back(one);
        vec.clear();
        cst = hf.zero();
        for (uint64_t pos = begin; pos < end; ++pos) {
            const H256 sc = st.back();
            st.pop_back();
            const uint8_t op = tok_op[pos];
            const uint32_t arg = tok_arg[pos];
            if (op == ACX_AFF_VAR) vec.emplace_back(flat(aff_wires[arg]), sc);
            else if (op == ACX_AFF_CONST) cst = hf.add(cst, sc == one ? scalars[arg] : cmul(sc, scalars[arg]));
            else if (op == ACX_AFF_SCALARMUL) st.push_back(sc == one ? scalars[arg] : cmul(sc, scalars[arg]));
            else { st.push_back(sc); st.push_back(sc); }
        }
        if (vec.size() > 1) {
            bool sorted = true;
            for (size_t i = 1; i < vec.size() && sorted; ++i) sorted = vec[i - 1].first < vec[i].first;
            if (!sorted) {
                sort_entries(vec);
                size_t w = 0;
                for (size_t i = 1; i < vec.size(); ++i) {
                    if (vec[i].first == vec[w].first) vec[w].second = hf.add(vec[w].second, vec[i].second);
                    else vec[++w] = vec[i];
                }
                vec.resize(w + 1);
            }
        }
    }

    // evalAffineCircuit of the tokens [begin, end): failed lookups are 0.  Same right-to-left evaluation.
    H256 affine_eval(uint64_t begin, uint64_t end, const std::vector<H256>& w, const std::vector<uint8_t>& assigned, const H256* scalars) const {
        std::vector<H256> st;                  // `scalars`: mont_scalars() -- the witness is kept in Montgomery form
        for (uint64_t pos = end; pos-- > begin;) {
            const uint8_t op = tok_op[pos];
            const uint32_t arg = tok_arg[pos];
            if (op == ACX_AFF_VAR) {
                const uint64_t k = flat(aff_wires[arg]);
                st.push_back(assigned[k] ? w[k] : hf.zero());
            } else if (op == ACX_AFF_CONST) {
                st.push_back(scalars[arg]);
            } else if (op == ACX_AFF_SCALARMUL) {
                st.back() = hf.mul(st.back(), scalars[arg]);
            } else {
                const H256 l = st.back();
                st.pop_back();
                st.back() = hf.add(l, st.back());
            }
        }
        return st.back();
    }

    // gateToGenQAP over every gate, rows in gate order.
    // gateToGenQAP for every gate (src/QAP.hs:366-474, 530-539).  Gates are independent: worker threads take contiguous gate
    // ranges, build their rows privately and copy them to their final offsets (rows stay in gate order).
    void build_rows(HostCsr& A, HostCsr& B, HostCsr& C) const {
        const unsigned T = host_threads(n_gates, 1 << 13);
        if (T <= 1) { build_rows_range(0, n_gates, A, B, C); return; }
        std::vector<HostCsr> part[3] = {std::vector<HostCsr>(T), std::vector<HostCsr>(T), std::vector<HostCsr>(T)};
        parallel_ranges(n_gates, T, [&](unsigned t, uint64_t gb, uint64_t ge) { build_rows_range(gb, ge, part[0][t], part[1][t], part[2][t]); });
        HostCsr* out[3] = {&A, &B, &C};
        std::vector<uint64_t> row0[3], nz0[3];
        for (int k = 0; k < 3; ++k) {
            row0[k].assign(T + 1, 0); nz0[k].assign(T + 1, 0);
            for (unsigned t = 0; t < T; ++t) {
                row0[k][t + 1] = row0[k][t] + part[k][t].rowptr.size() - 1;
                nz0[k][t + 1] = nz0[k][t] + part[k][t].col.size();
            }
            out[k]->rowptr.assign(row0[k][T] + 1, 0);
            out[k]->col.resize(nz0[k][T]);
            out[k]->val.resize(nz0[k][T]);
        }
        parallel_ranges(T, T, [&](unsigned t, uint64_t, uint64_t) {
            for (int k = 0; k < 3; ++k) {
                const HostCsr& src = part[k][t];
                std::copy(src.col.begin(), src.col.end(), out[k]->col.begin() + nz0[k][t]);
                std::copy(src.val.begin(), src.val.end(), out[k]->val.begin() + nz0[k][t]);
                for (size_t i = 1; i < src.rowptr.size(); ++i)
                    out[k]->rowptr[row0[k][t] + i] = (uint32_t)(nz0[k][t] + src.rowptr[i]);
            }
        });
    }

    // gateToGenQAP of ONE gate (src/QAP.hs:366-474): its rows as {A, B, C} entry lists (column, value) holding EVERY wire the
    // reference's row mentions -- the constant, the updated wires, explicit zeros included.  The zeros are void while roots are
    // distinct (push_row drops them); they decide what survives when two rows share a root (build_rows_reference).
    using RowMaps = std::array<Entries, 3>;
    // returns the gate's row count; rows[0 .. count) are filled (the vector only grows: its inner buffers are reused gate after gate)
    size_t gate_rows(uint64_t g, std::vector<RowMaps>& rows) const {
        const H256 one = c_one(), minus_one = hf.neg(c_one()), zero = hf.zero();      // canonical, like every row value
        const acx_wire* gw = &wires[wire_ofs[g]];
        const size_t count = rows_of_gate(g);
        if (rows.size() < count) rows.resize(count);
        for (size_t i = 0; i < count; ++i) for (auto& e : rows[i]) e.clear();
        if (kind[g] == ACX_GATE_MUL) {
            auto& l = rows[0][0]; auto& r = rows[0][1]; auto& o = rows[0][2];
            H256 lc, rc;
            affine_map(tok_ofs[2 * g], tok_ofs[2 * g + 1], lc, l);
            affine_map(tok_ofs[2 * g + 1], tok_ofs[2 * g + 2], rc, r);
            // constantQapSet (root, leftInputConst): a Var can never name column 0, so the constant goes in front
            l.insert(l.begin(), {0, lc});
            r.insert(r.begin(), {0, rc});
            o.emplace_back(0, zero);
            o.emplace_back(flat(gw[0]), one);
        } else if (kind[g] == ACX_GATE_EQUAL) {
            const uint64_t i = flat(gw[0]), mg = flat(gw[1]), out = flat(gw[2]);
            auto set3 = [&](H256 vi, H256 vm, H256 vo, H256 cst) {
                Entries row;
                row.emplace_back(0, cst);
                set_entry(row, i, vi); set_entry(row, mg, vm); set_entry(row, out, vo);  // updateAtWires: later entries overwrite
                sort_entries(row);
                return row;
            };
            rows[0] = {set3(one, zero, zero, zero), set3(zero, one, zero, zero), set3(zero, zero, one, zero)};      // row0: i * m = out
            rows[1] = {set3(zero, zero, minus_one, one), set3(one, zero, zero, zero), set3(zero, zero, zero, zero)}; // row1: (1 - out) * i = 0
        } else {
            const uint64_t n_outs = wire_ofs[g + 1] - wire_ofs[g] - 1;
            const uint64_t inp = flat(gw[0]);
            auto& a0 = rows[0][0]; auto& b0 = rows[0][1]; auto& c0 = rows[0][2];
            // updateAtWires ((inp, 0) : zip outputs 2^j) over the constant: a later pair for the same wire overwrites.  A std::map
            // here (one per Split gate): the overwrite needs a lookup and a row has 257 entries.
            std::map<uint64_t, H256> a0m;
            a0m[0] = zero; a0m[inp] = zero;
            H256 pw = one;  // 2^j
            for (uint64_t j = 0; j < n_outs; ++j) {
                a0m[flat(gw[1 + j])] = pw;
                pw = hf.add(pw, pw);
            }
            a0.assign(a0m.begin(), a0m.end());
            b0 = {{0, one}, {inp, zero}};
            c0 = {{0, zero}, {inp, one}};
            for (uint64_t j = 0; j < n_outs; ++j) {  // bit * (1 - bit) = 0
                const uint64_t o = flat(gw[1 + j]);
                rows[1 + j][0] = {{0, zero}, {o, one}};
                rows[1 + j][1] = {{0, one}, {o, minus_one}};
                rows[1 + j][2] = {{0, zero}, {o, zero}};
            }
        }
        return count;
    }

    void build_rows_range(uint64_t g_begin, uint64_t g_end, HostCsr& A, HostCsr& B, HostCsr& C) const {
        std::vector<RowMaps> rows;
        for (uint64_t g = g_begin; g < g_end; ++g) {
            const size_t count = gate_rows(g, rows);
            for (size_t i = 0; i < count; ++i) { A.push_row(hf, rows[i][0]); B.push_row(hf, rows[i][1]); C.push_row(hf, rows[i][2]); }
        }
    }

    // `arithCircuitToGenQAP rootsPerGate circuit` (src/QAP.hs:530-539) on ANY root lists, exactly as the reference computes it:
    //   zipWith gateToGenQAP rootsPerGate gates     lists beyond the last gate make no rows, gates beyond the last list are
    //                                               dropped; a list of the wrong length for its gate is the reference's panic
    //                                               (src/QAP.hs:444-445,474) = ACX_ERR_ROOT_COUNT
    //   createMapGenQap (src/QAP.hs:233-239)        per wire `Map.fromList` over the rows in order: of two rows with the SAME
    //                                               root the later one wins on every wire it mentions (explicit zeros and the
    //                                               constant included), wires it does not mention keep the earlier row's value
    //   addMissingZeroes (concat rootsPerGate)      every root of every list owns a row; roots no gate wrote are zero rows
    // roots: the lists concatenated (canonical); counts[g] = length of list g.  Out: one row per DISTINCT root, ascending
    // (`Map.elems`, src/QAP.hs:521-523), and those roots.  Sequential and map-based: degenerate lists are small.
    int build_rows_reference(const acx_fr* roots, const uint32_t* counts, uint64_t n_lists, HostCsr out[3], std::vector<H256>& distinct,
                             std::string& msg) const {
        const uint64_t used = std::min<uint64_t>(n_lists, n_gates);
        uint64_t total = 0;
        for (uint64_t g = 0; g < n_lists; ++g) {
            if (g < used && counts[g] != rows_of_gate(g)) { msg = "gateToGenQAP: wrong number of roots supplied"; return ACX_ERR_ROOT_COUNT; }
            total += counts[g];
        }
        if (total && !roots) { msg = "null root array"; return ACX_ERR_INVALID_ARG; }
        std::vector<H256> rv(total);
        for (uint64_t i = 0; i < total; ++i) {
            std::memcpy(rv[i].l, roots[i].b, 32);
            if (!hf.is_canonical(rv[i])) { msg = "root >= p"; return ACX_ERR_NONCANONICAL; }
        }
        auto less = [](const H256& a, const H256& b) { return h256_cmp(a, b) < 0; };
        distinct = rv;
        std::sort(distinct.begin(), distinct.end(), less);
        distinct.erase(std::unique(distinct.begin(), distinct.end()), distinct.end());
        std::vector<std::map<uint64_t, H256>> merged[3];
        for (auto& mk : merged) mk.resize(distinct.size());
        std::vector<RowMaps> rows;
        uint64_t pos = 0;
        for (uint64_t g = 0; g < used; ++g) {
            const size_t count = gate_rows(g, rows);
            for (size_t i = 0; i < count; ++i) {
                const uint64_t ri = (uint64_t)(std::lower_bound(distinct.begin(), distinct.end(), rv[pos++], less) - distinct.begin());
                for (int k = 0; k < 3; ++k)
                    for (const auto& kv : rows[i][k]) merged[k][ri][kv.first] = kv.second;       // Map.fromList: the later pair wins
            }
        }
        for (int k = 0; k < 3; ++k) {
            out[k] = HostCsr();
            for (const auto& row : merged[k]) out[k].push_row(hf, row);
        }
        return ACX_OK;
    }

    // generateAssignment.  inputs canonical.  Returns ACX_OK / ACX_ERR_UNDEFINED_WIRE.
    int eval(const acx_fr* inputs, const uint8_t* present, uint64_t n_inputs, std::vector<H256>& w,
             std::vector<uint8_t>& assigned, std::string& msg) const {
        const H256* sm = mont_scalars();
        w.assign(m(), hf.zero());
        assigned.assign(m(), 0);
        w[0] = hf.one();
        assigned[0] = 1;
        for (uint64_t i = 0; i < n_inputs && i < n_in; ++i) {
            if (present && !present[i]) continue;
            H256 c;
            std::memcpy(c.l, inputs[i].b, 32);
            if (!hf.is_canonical(c)) { msg = "input >= p"; return ACX_ERR_NONCANONICAL; }
            w[1 + i] = hf.to_mont(c);
            assigned[1 + i] = 1;
        }
        for (uint64_t g = 0; g < n_gates; ++g) {
            const acx_wire* gw = &wires[wire_ofs[g]];
            if (kind[g] == ACX_GATE_MUL) {
                const H256 l = affine_eval(tok_ofs[2 * g], tok_ofs[2 * g + 1], w, assigned, sm);
                const H256 r = affine_eval(tok_ofs[2 * g + 1], tok_ofs[2 * g + 2], w, assigned, sm);
                const uint64_t o = flat(gw[0]);
                w[o] = hf.mul(l, r);
                assigned[o] = 1;
            } else if (kind[g] == ACX_GATE_EQUAL) {
                const uint64_t i = flat(gw[0]), mg = flat(gw[1]), o = flat(gw[2]);
                if (!assigned[i]) { msg = "evalGate: the impossible happened (Equal input unassigned)"; return ACX_ERR_UNDEFINED_WIRE; }
                const bool z = w[i].is_zero();
                const H256 inv = z ? hf.zero() : hf.inv(w[i]);
                w[mg] = inv; assigned[mg] = 1;                       // updateVar m mid
                w[o] = z ? hf.zero() : hf.one(); assigned[o] = 1;    // then updateVar outputWire res
            } else {
                const uint64_t i = flat(gw[0]);
                if (!assigned[i]) { msg = "evalGate: the impossible happened (Split input unassigned)"; return ACX_ERR_UNDEFINED_WIRE; }
                const H256 c = hf.from_mont(w[i]);
                const uint64_t n_outs = wire_ofs[g + 1] - wire_ofs[g] - 1;
                for (uint64_t j = 0; j < n_outs; ++j) {
                    const bool bit = j < 256 && ((c.l[j / 64] >> (j % 64)) & 1);
                    const uint64_t o = flat(gw[1 + j]);
                    w[o] = bit ? hf.one() : hf.zero();
                    assigned[o] = 1;
                }
            }
        }
        return ACX_OK;
    }

    // Level schedule for evaluating the circuit on the GPU.  Possible when the
    // gate list is in single-assignment form: every intermediate/output wire is written by at most
    // one gate and only read by later gates (what `validArithCircuit` guarantees for reads; the
    // reference itself would let a later gate overwrite a wire -- such circuits keep the host path).
    struct EvalPlan {
        std::vector<uint32_t> level_ofs;   // [n_levels + 1] into items
        std::vector<uint32_t> items;       // gate ids, grouped by level
        std::vector<uint8_t> written;      // [m] wire is assigned by some gate (or is the constant)
        // Equal gates in gate order, when NO gate reads an Equal gate's magic wire (validArithCircuit never lets one: the magic
        // wire is not among outputWires, src/Circuit/Arithmetic.hs:158-185): the inversions then hang off the dependency graph
        // as leaves and are evaluated in one launch after the last level instead of adding their latency to every level.
        std::vector<uint32_t> deferred_equal;
        bool defer_magic = false;
    };
    bool build_plan(EvalPlan& plan) const {
        const uint64_t M = m();
        std::vector<int64_t> writer(M, -1);
        std::vector<uint8_t> is_magic(M, 0);
        bool magic_is_read = false;
        for (uint64_t g = 0; g < n_gates; ++g) {
            const acx_wire* gw = &wires[wire_ofs[g]];
            const uint64_t nw = wire_ofs[g + 1] - wire_ofs[g];
            auto claim = [&](const acx_wire& w) {
                const uint64_t k = flat(w);
                if (w.kind == ACX_WIRE_INPUT || writer[k] >= 0) return false;   // inputs are never gate outputs here
                writer[k] = (int64_t)g;
                return true;
            };
            if (kind[g] == ACX_GATE_MUL) { if (!claim(gw[0])) return false; }
            else if (kind[g] == ACX_GATE_EQUAL) { if (gw[1].kind == gw[2].kind && gw[1].index == gw[2].index) return false;
                                                   if (!claim(gw[1]) || !claim(gw[2])) return false;
                                                   is_magic[flat(gw[1])] = 1; }
            else for (uint64_t j = 1; j < nw; ++j) if (!claim(gw[j])) return false;
        }
        std::vector<uint32_t> level(n_gates, 0);
        uint32_t n_levels = 0;
        for (uint64_t g = 0; g < n_gates; ++g) {
            const acx_wire* gw = &wires[wire_ofs[g]];
            uint32_t lv = 0;
            bool ok = true;
            auto dep = [&](const acx_wire& w) {
                if (w.kind == ACX_WIRE_INPUT) return;
                const int64_t wr = writer[flat(w)];
                if (wr < 0) return;                        // never written: reads as absent
                if ((uint64_t)wr >= g) { ok = false; return; }   // written later: order-dependent
                if (is_magic[flat(w)]) magic_is_read = true;
                lv = std::max(lv, level[wr] + 1);
            };
            if (kind[g] == ACX_GATE_MUL) {
                for (uint64_t t = tok_ofs[2 * g]; t < tok_ofs[2 * g + 2]; ++t)
                    if (tok_op[t] == ACX_AFF_VAR) dep(aff_wires[tok_arg[t]]);
            } else dep(gw[0]);
            if (!ok) return false;
            level[g] = lv;
            n_levels = std::max(n_levels, lv + 1);
        }
        plan.level_ofs.assign(n_levels + 1, 0);
        for (uint64_t g = 0; g < n_gates; ++g) ++plan.level_ofs[level[g] + 1];
        for (uint32_t l = 0; l < n_levels; ++l) plan.level_ofs[l + 1] += plan.level_ofs[l];
        plan.items.resize(n_gates);
        std::vector<uint32_t> cur(plan.level_ofs.begin(), plan.level_ofs.end() - 1);
        for (uint64_t g = 0; g < n_gates; ++g) plan.items[cur[level[g]]++] = (uint32_t)g;
        plan.written.assign(M, 0);
        plan.written[0] = 1;
        for (uint64_t k = 0; k < M; ++k) if (writer[k] >= 0) plan.written[k] = 1;
        plan.defer_magic = !magic_is_read;
        plan.deferred_equal.clear();
        if (plan.defer_magic)
            for (uint64_t g = 0; g < n_gates; ++g) if (kind[g] == ACX_GATE_EQUAL) plan.deferred_equal.push_back((uint32_t)g);
        return true;
    }

    bool valid() const {
        std::vector<uint8_t> defined(n_mid, 0);
        bool res = true;
        for (uint64_t g = 0; g < n_gates; ++g) {
            const acx_wire* gw = &wires[wire_ofs[g]];
            const uint64_t nw = wire_ofs[g + 1] - wire_ofs[g];
            auto valid_wire = [&](const acx_wire& w) {
                if (w.kind == ACX_WIRE_INPUT) return true;
                if (w.kind == ACX_WIRE_OUTPUT) return false;
                return defined[w.index] != 0;
            };
            bool ok = true;
            if (kind[g] == ACX_GATE_MUL) {
                ok = gw[0].kind != ACX_WIRE_INPUT;
                for (uint64_t t = tok_ofs[2 * g]; t < tok_ofs[2 * g + 2]; ++t)
                    if (tok_op[t] == ACX_AFF_VAR) ok = ok && valid_wire(aff_wires[tok_arg[t]]);
            } else if (kind[g] == ACX_GATE_EQUAL) {
                ok = gw[2].kind != ACX_WIRE_INPUT && valid_wire(gw[0]);  // outputWires = [eqOutput]
            } else {
                for (uint64_t j = 1; j < nw; ++j) ok = ok && gw[j].kind != ACX_WIRE_INPUT;
                ok = ok && valid_wire(gw[0]);
            }
            res = res && ok;
            // outputWires gate ++ definedWires
            if (kind[g] == ACX_GATE_MUL) { if (gw[0].kind == ACX_WIRE_INTERMEDIATE) defined[gw[0].index] = 1; }
            else if (kind[g] == ACX_GATE_EQUAL) { if (gw[2].kind == ACX_WIRE_INTERMEDIATE) defined[gw[2].index] = 1; }
            else for (uint64_t j = 1; j < nw; ++j) if (gw[j].kind == ACX_WIRE_INTERMEDIATE) defined[gw[j].index] = 1;
        }
        return res;
    }

private:
    // One well-formed pre-order tree starting at pos (advanced past it): every token opens as many
    // sub-trees as its arity; the tree is complete when none is left open.  No recursion.
    bool check_tree(uint64_t& pos, uint64_t end, const acx_gate_list* gl) const {
        uint64_t open = 1;
        while (open) {
            if (pos >= end) return false;
            const uint8_t op = tok_op[pos];
            const uint32_t arg = tok_arg[pos];
            ++pos;
            --open;
            switch (op) {
                case ACX_AFF_VAR: if (arg >= gl->n_aff_wires) return false; break;
                case ACX_AFF_CONST: if (arg >= gl->n_scalars) return false; break;
                case ACX_AFF_SCALARMUL: if (arg >= gl->n_scalars) return false; open += 1; break;
                case ACX_AFF_ADD: open += 2; break;
                default: return false;
            }
        }
        return true;
    }
};

}  // namespace acx
