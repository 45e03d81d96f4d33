// circuit_abi.inc.h -- the entry points of include/acx.h that are PURE HOST CODE: error text, version, and everything on
// acx_circuit (marshalling, validation, gateToGenQAP rows, generateAssignment).  Included inside the extern "C" block of
// engine.hip (libacx.so) and of host_only.cpp (the same symbols in a library with no HIP in it, built with
// -fsanitize=address,undefined: the gate list is an untrusted token stream).
const char* acx_strerror(int status) {
    switch (status) {
        case ACX_OK: return "ok";
        case ACX_ERR_INVALID_ARG: return "invalid argument";
        case ACX_ERR_NONCANONICAL: return "field element is not canonical (>= p)";
        case ACX_ERR_NO_DEVICE: return "no usable HIP device";
        case ACX_ERR_HIP: return "HIP runtime error";
        case ACX_ERR_ROOT_COUNT: return "gateToGenQAP: wrong number of roots supplied";
        case ACX_ERR_UNDEFINED_WIRE: return "evalGate: the impossible happened (undefined wire)";
        case ACX_ERR_DUPLICATE_ROOT: return "duplicate root";
        case ACX_ERR_TOO_LARGE: return "size exceeds supported range";
        case ACX_ERR_OOM: return "out of memory";
        case ACX_ERR_BAD_CIRCUIT: return "malformed marshalled circuit";
        case ACX_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

const char* acx_last_error(void) { return g_last_error.c_str(); }
uint32_t acx_version(void) { return ACX_VERSION; }

// ---------------------------------------------------------------------------------- circuit
int acx_circuit_create(int field, const acx_gate_list* gates, acx_circuit** out) {
    if (!gates || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    if (field != ACX_FIELD_BN254_FR && field != ACX_FIELD_BLS12_381_FR) return fail(ACX_ERR_INVALID_ARG, "unknown field");
    return guarded([&]() -> int {
        std::unique_ptr<acx_circuit> c(new acx_circuit());
        c->field = field;
        c->hc.hf = field == ACX_FIELD_BN254_FR ? HostField::make<Bn254Fr>() : HostField::make<Bls12381Fr>();
        std::string msg;
        PhaseTimer pt;
        const int rc = c->hc.init(gates, msg);
        if (rc != ACX_OK) return fail(rc, msg);
        pt.mark("circuit: copy + validate");
        c->hc.build_rows(c->rows[0], c->rows[1], c->rows[2]);
        pt.mark("circuit: gateToGenQAP rows");
        *out = c.release();
        return ACX_OK;
    });
}

int acx_circuit_check_root_counts(const acx_circuit* c, const uint32_t* counts, uint64_t n_lists) {
    if (!c || (n_lists && !counts)) return fail(ACX_ERR_INVALID_ARG, "null argument");
    // `zipWith gateToGenQAP rootsPerGate gates` (src/QAP.hs:539): list g must hold exactly the gate's row count
    // (src/QAP.hs:444-445,474 panic otherwise); lists beyond the last gate are a deviation documented in acx.h
    if (n_lists != c->hc.n_gates) return fail(ACX_ERR_ROOT_COUNT, "gateToGenQAP: one root list per gate is required");
    for (uint64_t g = 0; g < n_lists; ++g)
        if (counts[g] != c->hc.rows_of_gate(g)) return fail(ACX_ERR_ROOT_COUNT, "gateToGenQAP: wrong number of roots supplied");
    return ACX_OK;
}

void acx_circuit_destroy(acx_circuit* c) {
    if (!c) return;
    for (auto& m : c->rows) { HostCsr empty; std::swap(m, empty); }      // the rows are never needed by a pending plan
    circuit_release(c);
}

int acx_circuit_dims(const acx_circuit* c, uint64_t* n_rows, uint64_t* m_wires, uint64_t* n_inputs,
                     uint64_t* n_intermediates, uint64_t* n_outputs) {
    if (!c) return fail(ACX_ERR_INVALID_ARG, "null circuit");
    if (n_rows) *n_rows = c->hc.n_rows();
    if (m_wires) *m_wires = c->hc.m();
    if (n_inputs) *n_inputs = c->hc.n_in;
    if (n_intermediates) *n_intermediates = c->hc.n_mid;
    if (n_outputs) *n_outputs = c->hc.n_out;
    return ACX_OK;
}

int acx_circuit_rows_per_gate(const acx_circuit* c, uint32_t* out) {
    if (!c || !out) return fail(ACX_ERR_INVALID_ARG, "null argument");
    for (uint64_t g = 0; g < c->hc.n_gates; ++g) out[g] = (uint32_t)c->hc.rows_of_gate(g);
    return ACX_OK;
}

int acx_circuit_valid(const acx_circuit* c, int* valid) {
    if (!c || !valid) return fail(ACX_ERR_INVALID_ARG, "null argument");
    return guarded([&]() -> int {          // the scan allocates per-wire state: a huge wire index must be an error code, not a throw
        *valid = c->hc.valid() ? 1 : 0;
        return ACX_OK;
    });
}

int acx_circuit_eval(const acx_circuit* c, const acx_fr* inputs, const uint8_t* present, uint64_t n_inputs,
                     acx_fr* witness, uint8_t* assigned) {
    if (!c || !witness || (n_inputs && !inputs)) return fail(ACX_ERR_INVALID_ARG, "null argument");
    return guarded([&]() -> int {
        std::vector<H256> w;
        std::vector<uint8_t> as;
        std::string msg;
        const int rc = c->hc.eval(inputs, present, n_inputs, w, as, msg);
        if (rc != ACX_OK) return fail(rc, msg);
        for (uint64_t k = 0; k < w.size(); ++k) write_h256(&witness[k], c->hc.hf, w[k]);
        if (assigned) std::memcpy(assigned, as.data(), as.size());
        return ACX_OK;
    });
}

// rows in ascending-root order (`Map.elems`, src/QAP.hs:521-523); empty order = identity
static int root_order(const HostCircuit& hc, const acx_fr* roots, uint64_t n_roots, std::vector<uint64_t>& order) {
    const uint64_t n = hc.n_rows();
    order.clear();
    if (!roots) return ACX_OK;
    if (n_roots != n) return fail(ACX_ERR_ROOT_COUNT, "gateToGenQAP: wrong number of roots supplied");
    std::vector<H256> rv(n);
    for (uint64_t i = 0; i < n; ++i) {
        std::memcpy(rv[i].l, roots[i].b, 32);
        if (!hc.hf.is_canonical(rv[i])) return fail(ACX_ERR_NONCANONICAL, "root >= p");
    }
    order.resize(n);
    for (uint64_t i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint64_t a, uint64_t b) { return h256_cmp(rv[a], rv[b]) < 0; });
    bool identity = true;
    for (uint64_t i = 0; i < n; ++i) {
        if (i && rv[order[i]] == rv[order[i - 1]]) return fail(ACX_ERR_DUPLICATE_ROOT, "roots must be distinct");
        identity = identity && order[i] == i;
    }
    if (identity) order.clear();
    return ACX_OK;
}

static void permute_rows(const HostCsr& src, const std::vector<uint64_t>& order, HostCsr& dst) {
    dst = HostCsr();
    for (uint64_t s : order) {
        for (uint32_t e = src.rowptr[s]; e < src.rowptr[s + 1]; ++e) {
            dst.col.push_back(src.col[e]);
            dst.val.push_back(src.val[e]);
        }
        dst.rowptr.push_back((uint32_t)dst.col.size());
    }
}

int acx_circuit_nnz(const acx_circuit* c, uint64_t nnz[3]) {
    if (!c || !nnz) return fail(ACX_ERR_INVALID_ARG, "null argument");
    for (int k = 0; k < 3; ++k) nnz[k] = c->rows[k].col.size();
    return ACX_OK;
}

int acx_circuit_rows(const acx_circuit* c, const acx_fr* roots, uint64_t n_roots, int matrix, uint32_t* rowptr,
                     uint32_t* col, acx_fr* val) {
    if (!c || matrix < 0 || matrix > 2 || !rowptr) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    return guarded([&]() -> int {
        std::vector<uint64_t> order;
        ACX_TRY(root_order(c->hc, roots, n_roots, order));
        HostCsr perm;
        const HostCsr* src = &c->rows[matrix];
        if (!order.empty()) { permute_rows(*src, order, perm); src = &perm; }
        std::memcpy(rowptr, src->rowptr.data(), src->rowptr.size() * 4);
        if (col && !src->col.empty()) std::memcpy(col, src->col.data(), src->col.size() * 4);
        if (val && !src->val.empty()) std::memcpy(val, src->val.data(), src->val.size() * 32);
        return ACX_OK;
    });
}

// `arithCircuitToGenQAP rootsPerGate circuit` with the roots as the reference takes them, one list PER GATE.
// Lists that are one-per-gate, of the right lengths and pairwise distinct are the ordinary case: the flat call.  Anything else
// is an error without ACX_ROOTS_REFERENCE_SEMANTICS and the reference's own result with it (HostCircuit::build_rows_reference).
static int lists_are_regular(const HostCircuit& hc, const acx_fr* roots, const uint32_t* counts, uint64_t n_lists, bool* regular) {
    *regular = false;
    if (n_lists != hc.n_gates) return ACX_OK;
    uint64_t total = 0;
    for (uint64_t g = 0; g < n_lists; ++g) {
        if (counts[g] != hc.rows_of_gate(g)) return ACX_OK;
        total += counts[g];
    }
    if (total && !roots) return fail(ACX_ERR_INVALID_ARG, "null root array");
    std::vector<H256> rv(total);
    for (uint64_t i = 0; i < total; ++i) std::memcpy(rv[i].l, roots[i].b, 32);
    std::sort(rv.begin(), rv.end(), [](const H256& a, const H256& b) { return h256_cmp(a, b) < 0; });
    *regular = std::adjacent_find(rv.begin(), rv.end()) == rv.end();
    return ACX_OK;
}

int acx_circuit_rows_lists(const acx_circuit* c, const acx_fr* roots, const uint32_t* counts, uint64_t n_lists, uint32_t flags, int matrix,
                           uint64_t* n_rows, uint64_t* nnz, uint32_t* rowptr, uint32_t* col, acx_fr* val, acx_fr* sorted_roots) {
    if (!c || matrix < 0 || matrix > 2 || (n_lists && !counts)) return fail(ACX_ERR_INVALID_ARG, "bad argument");
    if (flags & ~(uint32_t)ACX_ROOTS_REFERENCE_SEMANTICS) return fail(ACX_ERR_INVALID_ARG, "unknown flag");
    return guarded([&]() -> int {
        if (!(flags & ACX_ROOTS_REFERENCE_SEMANTICS)) {
            ACX_TRY(acx_circuit_check_root_counts(c, counts, n_lists));
            uint64_t total = 0;
            for (uint64_t g = 0; g < n_lists; ++g) total += counts[g];
            std::vector<uint64_t> order;
            ACX_TRY(root_order(c->hc, roots, total, order));       // ACX_ERR_DUPLICATE_ROOT on a repeated root
        }
        HostCsr M[3];
        std::vector<H256> distinct;
        std::string msg;
        const int rc = c->hc.build_rows_reference(roots, counts, n_lists, M, distinct, msg);
        if (rc != ACX_OK) return fail(rc, msg);
        const HostCsr& src = M[matrix];
        if (n_rows) *n_rows = distinct.size();
        if (nnz) *nnz = src.col.size();
        if (rowptr) std::memcpy(rowptr, src.rowptr.data(), src.rowptr.size() * 4);
        if (col && !src.col.empty()) std::memcpy(col, src.col.data(), src.col.size() * 4);
        if (val && !src.val.empty()) std::memcpy(val, src.val.data(), src.val.size() * 32);
        if (sorted_roots && !distinct.empty()) std::memcpy(sorted_roots, distinct.data(), distinct.size() * 32);
        return ACX_OK;
    });
}

