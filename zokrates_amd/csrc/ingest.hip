// ingest.hip — host-side readers of ZoKrates' own files ("next" row N1 of SURVEY.md §8f; rows a2, a3, a5, a6 of §8a).
//
//   `out`      ProgEnum::deserialize            /root/reference/zokrates_ast/src/ir/serialize.rs:133-199 (header),
//                                               :306-390 (sections, lazy per-statement CBOR decode)
//              serde derive shapes              zokrates_ast/src/ir/mod.rs:34-41 (ConstraintStatement), :119-128
//                                               (Statement), ir/expression.rs:10-18,69-76 (QuadComb, LinComb),
//                                               common/flat/variable.rs:10-13, common/flat/parameter.rs:9-16,
//                                               field elements as CBOR byte strings zokrates_field/src/lib.rs:547-560
//   ark order  Computation::generate_constraints /root/reference/zokrates_ark/src/lib.rs:41-130
//   `witness`  Witness::read                    /root/reference/zokrates_ast/src/ir/witness.rs:55-71
//   inputs     ProgIterator::public_inputs_values /root/reference/zokrates_ast/src/ir/mod.rs:278-288
//
// The CBOR is what [UPSTREAM] serde_cbor 0.11.2 (`Cargo.lock:2750`) writes for derived `Serialize`: structs are
// maps keyed by field name, enums are externally tagged (`{"Variant": value}`, unit variants a bare string),
// `None` is null, sequences are arrays, integers take the shortest form.  Nothing but constraints is decoded:
// directives, logs, spans and error annotations are skipped structurally.
//
// Pure host code: no kernel, no device memory.  This translation unit is C++ that hipcc compiles like the others.
#include "ingest.h"

#include <cstring>

#include "devrt.h"
#include "field.cuh"
#include "../../include/zkhip.h"

namespace zk {
namespace {

[[noreturn]] void fail(int32_t code, const std::string& msg) { throw IngestError{code, msg}; }

// ------------------------------------------------------------------ minimal CBOR pull reader (RFC 8949 subset)
struct Cbor {
    const uint8_t* p;
    const uint8_t* e;
    struct Head {
        int major;
        uint64_t arg;
        bool indef;
    };
    static constexpr uint64_t INDEF = ~(uint64_t)0;

    bool at_end() const { return p >= e; }
    uint8_t peek() const {
        if (p >= e) fail(ZKHIP_ERR_PARSE, "program file truncated (CBOR item expected)");
        return *p;
    }
    const uint8_t* take(uint64_t n) {
        if ((uint64_t)(e - p) < n) fail(ZKHIP_ERR_PARSE, "program file truncated inside a CBOR item");
        const uint8_t* r = p;
        p += n;
        return r;
    }
    Head head() {
        const uint8_t b = *take(1);
        Head h{b >> 5, 0, false};
        const int ai = b & 31;
        if (ai < 24) h.arg = ai;
        else if (ai <= 27) {
            const int nb = 1 << (ai - 24);
            const uint8_t* q = take(nb);
            for (int i = 0; i < nb; ++i) h.arg = (h.arg << 8) | q[i];
        } else if (ai == 31) h.indef = true;
        else fail(ZKHIP_ERR_PARSE, "malformed CBOR (reserved additional information)");
        return h;
    }
    bool is_break() const { return p < e && *p == 0xff; }
    void skip(int depth = 0) {
        if (depth > 64) fail(ZKHIP_ERR_PARSE, "CBOR nesting too deep");
        const Head h = head();
        switch (h.major) {
            case 0: case 1: return;
            case 2: case 3:
                if (!h.indef) { take(h.arg); return; }
                while (!is_break()) skip(depth + 1);
                take(1);
                return;
            case 4: case 5: {
                const uint64_t per = h.major == 5 ? 2 : 1;
                if (!h.indef) {
                    for (uint64_t i = 0; i < h.arg * per; ++i) skip(depth + 1);
                    return;
                }
                while (!is_break()) skip(depth + 1);
                take(1);
                return;
            }
            case 6: skip(depth + 1); return;
            default:   // 7: simple values and floats carry no payload beyond the argument; "break" outside a container is an error
                if (h.indef) fail(ZKHIP_ERR_PARSE, "malformed CBOR (unexpected break)");
                return;
        }
    }
    // containers: returns the element count, or INDEF for indefinite length (then poll `more`)
    uint64_t enter(int major, const char* what) {
        const Head h = head();
        if (h.major != major) fail(ZKHIP_ERR_PARSE, std::string("unexpected CBOR type where ") + what + " was expected");
        return h.indef ? INDEF : h.arg;
    }
    bool more(uint64_t& remaining) {
        if (remaining == INDEF) {
            if (is_break()) { take(1); return false; }
            return true;
        }
        if (remaining == 0) return false;
        --remaining;
        return true;
    }
    // text key (definite length only, which is all serde_cbor writes for identifiers)
    std::string text(const char* what) {
        const Head h = head();
        if (h.major != 3 || h.indef) fail(ZKHIP_ERR_PARSE, std::string("text string expected for ") + what);
        const uint8_t* q = take(h.arg);
        return std::string((const char*)q, (size_t)h.arg);
    }
    int64_t integer(const char* what) {
        const Head h = head();
        if (h.major == 0 && h.arg <= (uint64_t)INT64_MAX) return (int64_t)h.arg;
        if (h.major == 1 && h.arg <= (uint64_t)INT64_MAX) return -1 - (int64_t)h.arg;
        fail(ZKHIP_ERR_PARSE, std::string("integer expected for ") + what);
    }
    bool boolean(const char* what) {
        const uint8_t b = *take(1);
        if (b == 0xf4) return false;
        if (b == 0xf5) return true;
        fail(ZKHIP_ERR_PARSE, std::string("boolean expected for ") + what);
    }
};

// ------------------------------------------------------------------ ark variable allocation (lib.rs:80-129)
constexpr uint32_t WIT_TAG = 0x80000000u;   // during the walk: instance index, or WIT_TAG | witness index
constexpr uint32_t UNSEEN = 0xffffffffu;

struct Symbols {
    std::vector<uint32_t> pos, neg;   // tag by ZoKrates id: pos[id] for id >= 0, neg[-id - 1] for id < 0
    std::vector<int64_t> inst, wit;   // ids in allocation order (inst[0] = ~one)
    uint64_t id_limit = (uint64_t)1 << 31;   // the compiler numbers variables densely: an id far beyond what the file could
                                             // define is corruption, not a reason to allocate gigabytes of lookup table
    uint32_t& slot(int64_t id) {
        std::vector<uint32_t>& v = id >= 0 ? pos : neg;
        const uint64_t k = id >= 0 ? (uint64_t)id : (uint64_t)(-(id + 1));
        if (k >= id_limit) fail(ZKHIP_ERR_PARSE, "variable id out of range");
        if (k >= v.size()) v.resize(std::max<size_t>(k + 1, v.size() * 2), UNSEEN);
        return v[k];
    }
    uint32_t alloc(int64_t id, bool instance) {
        std::vector<int64_t>& dst = instance ? inst : wit;
        if (dst.size() >= (WIT_TAG >> 1)) fail(ZKHIP_ERR_BAD_ARG, "too many variables");
        const uint32_t tag = (instance ? 0u : WIT_TAG) | (uint32_t)dst.size();
        dst.push_back(id);
        return tag;
    }
    // symbols.entry(k).or_insert_with(new_input_variable | new_witness_variable)   lib.rs:50-70
    uint32_t lookup(int64_t id) {
        uint32_t& s = slot(id);
        if (s == UNSEEN) s = alloc(id, id < 0);
        return s;
    }
};

template <class P>
struct Builder {
    typedef Fe<P> Fr;
    struct Term {
        uint32_t tag;
        Fr coeff;
    };
    Symbols sym;
    std::vector<Term> rows[3];          // all terms, tags unresolved
    std::vector<uint64_t> rp[3];
    std::vector<Term> scratch;

    static bool canonical(const Fr& x) {
        for (int i = P::N - 1; i >= 0; --i)
            if (x.v[i] != P::mod(i)) return x.v[i] < P::mod(i);
        return false;
    }
    // LinComb {span, value: [[{id}, bytes], ...]} -> one matrix row: duplicates summed (ark's `acc + (coeff, var)`
    // merges equal variables), zero coefficients dropped (ark drops them when the matrices are extracted)
    void lin_comb(Cbor& c, int which) {
        scratch.clear();
        uint64_t nf = c.enter(5, "a linear combination");
        bool have_value = false;
        while (c.more(nf)) {
            const std::string key = c.text("a LinComb field");
            if (key != "value") { c.skip(); continue; }
            have_value = true;
            uint64_t nt = c.enter(4, "LinComb.value");
            while (c.more(nt)) {
                uint64_t pair = c.enter(4, "a (variable, coefficient) pair");
                if (pair != 2) fail(ZKHIP_ERR_PARSE, "LinComb term is not a pair");
                int64_t id = 0;
                bool have_id = false;
                uint64_t vf = c.enter(5, "a variable");
                while (c.more(vf)) {
                    if (c.text("a Variable field") == "id") { id = c.integer("Variable.id"); have_id = true; }
                    else c.skip();
                }
                if (!have_id) fail(ZKHIP_ERR_PARSE, "Variable without id");
                const Cbor::Head h = c.head();
                if (h.major != 2 || h.indef || h.arg != 32) fail(ZKHIP_ERR_PARSE, "field element is not a 32-byte string");
                Term t;
                memcpy(t.coeff.v, c.take(32), 32);
                if (!canonical(t.coeff)) fail(ZKHIP_ERR_PARSE, "non-canonical field element in the program");
                t.tag = sym.lookup(id);
                scratch.push_back(t);
            }
        }
        if (!have_value) fail(ZKHIP_ERR_PARSE, "LinComb without value");
        // merge duplicates; rows are tiny, so a quadratic pass beats sorting
        for (size_t i = 0; i < scratch.size(); ++i) {
            if (scratch[i].tag == UNSEEN) continue;
            for (size_t j = i + 1; j < scratch.size(); ++j)
                if (scratch[j].tag == scratch[i].tag) {
                    scratch[i].coeff = fe_add(scratch[i].coeff, scratch[j].coeff);
                    scratch[j].tag = UNSEEN;
                }
            if (!scratch[i].coeff.is_zero()) rows[which].push_back(scratch[i]);
        }
        rp[which].push_back(rows[which].size());
    }
    void constraint(Cbor& c) {
        uint64_t nf = c.enter(5, "a constraint statement");
        int seen = 0;
        // serde writes the fields in declaration order (span, quad, lin, error): quad before lin, which is the order
        // ark_combination is called in (left, right, lin) and hence the order variables are allocated in
        while (c.more(nf)) {
            const std::string key = c.text("a ConstraintStatement field");
            if (key == "quad") {
                if (seen != 0) fail(ZKHIP_ERR_PARSE, "ConstraintStatement fields out of order");
                uint64_t qf = c.enter(5, "a quadratic combination");
                int qseen = 0;
                while (c.more(qf)) {
                    const std::string qk = c.text("a QuadComb field");
                    if (qk == "left") { if (qseen != 0) fail(ZKHIP_ERR_PARSE, "QuadComb fields out of order"); lin_comb(c, 0); qseen = 1; }
                    else if (qk == "right") { if (qseen != 1) fail(ZKHIP_ERR_PARSE, "QuadComb fields out of order"); lin_comb(c, 1); qseen = 2; }
                    else c.skip();
                }
                if (qseen != 2) fail(ZKHIP_ERR_PARSE, "QuadComb without left/right");
                seen = 1;
            } else if (key == "lin") {
                if (seen != 1) fail(ZKHIP_ERR_PARSE, "ConstraintStatement fields out of order");
                lin_comb(c, 2);
                seen = 2;
            } else {
                c.skip();
            }
        }
        if (seen != 2) fail(ZKHIP_ERR_PARSE, "ConstraintStatement without quad/lin");
    }
    void run(const uint8_t* bytes, uint64_t par_off, uint64_t par_len, uint64_t st_off, uint64_t st_len, zkhip_prog* out) {
        for (auto& r : rp) r.assign(1, 0);
        sym.id_limit = std::min<uint64_t>((uint64_t)1 << 31, std::max<uint64_t>((uint64_t)1 << 20, 8 * (par_len + st_len)));
        // symbols[~one] = ConstraintSystem::one()                                   lib.rs:89
        sym.slot(0) = sym.alloc(0, true);
        // arguments, in order: private -> witness, public -> instance               lib.rs:94-113
        {
            Cbor c{bytes + par_off, bytes + par_off + par_len};
            uint64_t np = c.enter(4, "the parameter list");
            while (c.more(np)) {
                uint64_t nf = c.enter(5, "a parameter");
                int64_t id = 0;
                bool priv = false, have_id = false, have_priv = false;
                while (c.more(nf)) {
                    const std::string key = c.text("a Parameter field");
                    if (key == "id") {
                        uint64_t vf = c.enter(5, "a variable");
                        while (c.more(vf)) {
                            if (c.text("a Variable field") == "id") { id = c.integer("Variable.id"); have_id = true; }
                            else c.skip();
                        }
                    } else if (key == "private") {
                        priv = c.boolean("Parameter.private");
                        have_priv = true;
                    } else {
                        c.skip();
                    }
                }
                if (!have_id || !have_priv) fail(ZKHIP_ERR_PARSE, "Parameter without id/private");
                sym.slot(id) = sym.alloc(id, !priv);   // `symbols.extend`: a repeated parameter re-binds the name
                if (!priv) out->public_args.push_back(id);
            }
        }
        // statements: a stream of CBOR items, constraints only                      lib.rs:115-123
        {
            Cbor c{bytes + st_off, bytes + st_off + st_len};
            while (!c.at_end()) {
                const uint8_t b = c.peek();
                if ((b >> 5) == 3) { c.skip(); continue; }          // a unit variant would be a bare string: none of ours
                uint64_t one = c.enter(5, "a statement");
                if (one != 1) fail(ZKHIP_ERR_PARSE, "statement is not a single-entry map");
                const std::string variant = c.text("the statement variant");
                if (variant == "Constraint") constraint(c);
                else c.skip();                                      // Directive, Log: witness generation only
            }
        }
        const uint64_t l = sym.inst.size(), w = sym.wit.size(), n = rp[0].size() - 1;
        if (l + w + 2 >= ((uint64_t)1 << 31)) fail(ZKHIP_ERR_BAD_ARG, "too many variables");
        out->n = n; out->l = l; out->w = w;
        out->order = sym.inst;
        out->order.insert(out->order.end(), sym.wit.begin(), sym.wit.end());
        for (int k = 0; k < 3; ++k) {
            const size_t nnz = rows[k].size();
            out->rp[k] = rp[k];
            out->col[k].resize(nnz);
            out->val[k].resize(nnz * 32);
            for (size_t q = 0; q < nnz; ++q) {
                const uint32_t t = rows[k][q].tag;
                out->col[k][q] = (t & WIT_TAG) ? (uint32_t)(l + (t & ~WIT_TAG)) : t;
                memcpy(&out->val[k][q * 32], rows[k][q].coeff.v, 32);
            }
            // ark keeps a row sorted by variable (One, Instance(i), Witness(i)): same order as the final column index
            for (uint64_t i = 0; i < n; ++i) {
                const uint64_t a = rp[k][i], b = rp[k][i + 1];
                for (uint64_t x = a + 1; x < b; ++x)          // insertion sort, rows are tiny
                    for (uint64_t y = x; y > a && out->col[k][y - 1] > out->col[k][y]; --y) {
                        std::swap(out->col[k][y - 1], out->col[k][y]);
                        uint8_t tmp[32];
                        memcpy(tmp, &out->val[k][(y - 1) * 32], 32);
                        memcpy(&out->val[k][(y - 1) * 32], &out->val[k][y * 32], 32);
                        memcpy(&out->val[k][y * 32], tmp, 32);
                    }
            }
            rows[k].clear();
            rows[k].shrink_to_fit();
        }
    }
};

uint32_t rd32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }
uint64_t rd64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

template <class P>
bool canonical32(const uint8_t* b) {
    Fe<P> x;
    memcpy(x.v, b, 32);
    return Builder<P>::canonical(x);
}

}  // namespace

// curve id = first 4 bytes of sha256(modulus, little-endian bytes) (/root/reference/zokrates_field/src/lib.rs:283-293);
// bn128's value is pinned by /root/reference/zokrates_book/src/toolbox/ir.md:15 (b4f7b5bd), bls12_381's recomputed (40d8c1f9)
static const uint8_t ID_BN128[4] = {0xb4, 0xf7, 0xb5, 0xbd};
static const uint8_t ID_BLS12_381[4] = {0x40, 0xd8, 0xc1, 0xf9};

void prog_parse(const uint8_t* bytes, size_t len, zkhip_prog* out) {
    // ProgHeader::read (serialize.rs:150-199): magic, version, curve id, constraint count, return count, 4 x (type u32, offset u64, length u64)
    constexpr size_t HEADER = 20 + 4 * 20;
    if (len < HEADER) fail(ZKHIP_ERR_PARSE, "Invalid header");
    static const uint8_t MAGIC[4] = {0x5a, 0x4f, 0x4b, 0}, VERSION[4] = {3, 0, 0, 0};
    if (memcmp(bytes, MAGIC, 4)) fail(ZKHIP_ERR_PARSE, "Invalid magic number");
    if (memcmp(bytes + 4, VERSION, 4)) fail(ZKHIP_ERR_PARSE, "Invalid file version");
    if (!memcmp(bytes + 8, ID_BN128, 4)) out->curve = ZKHIP_CURVE_BN128;
    else if (!memcmp(bytes + 8, ID_BLS12_381, 4)) out->curve = ZKHIP_CURVE_BLS12_381;
    else fail(ZKHIP_ERR_BAD_ARG, "Unknown curve identifier (this backend proves over bn128 and bls12_381)");
    const uint32_t constraint_count = rd32(bytes + 12);
    out->return_count = rd32(bytes + 16);
    uint64_t off[4], ln[4];
    for (int s = 0; s < 4; ++s) {
        const uint8_t* q = bytes + 20 + 20 * s;
        const uint32_t ty = rd32(q);
        if (ty < 1 || ty > 4) fail(ZKHIP_ERR_PARSE, "invalid section type");
        off[s] = rd64(q + 4);
        ln[s] = rd64(q + 12);
        if (off[s] > len || ln[s] > len - off[s]) fail(ZKHIP_ERR_PARSE, "section outside the file");
    }
    if (out->curve == ZKHIP_CURVE_BN128) {
        Builder<Bn254Fr> b;
        b.run(bytes, off[0], ln[0], off[1], ln[1], out);
    } else {
        Builder<Bls381Fr> b;
        b.run(bytes, off[0], ln[0], off[1], ln[1], out);
    }
    if (out->n != constraint_count) fail(ZKHIP_ERR_PARSE, "constraint count in the header does not match the statements");
}

void prog_assignment(const zkhip_prog* prog, const uint8_t* wit, size_t len, uint8_t* z_out, uint8_t* inputs_out, uint64_t cap, uint64_t* n_inputs) {
    // Witness::read: usize count, then (isize id, 32-byte canonical LE value) per entry          witness.rs:55-71
    if (len < 8) fail(ZKHIP_ERR_PARSE, "witness file truncated");
    const uint64_t count = rd64(wit);
    if (count > (len - 8) / 40 || len != 8 + 40 * count) fail(ZKHIP_ERR_PARSE, "witness file size does not match its entry count");
    std::vector<uint32_t> pos, neg;   // entry index + 1 by id (a BTreeMap insert: the last duplicate wins)
    uint64_t n_out = 0;
    for (uint64_t i = 0; i < count; ++i) {
        const uint8_t* ent = wit + 8 + 40 * i;
        int64_t id;
        memcpy(&id, ent, 8);
        const bool ok = prog->curve == ZKHIP_CURVE_BN128 ? canonical32<Bn254Fr>(ent + 8) : canonical32<Bls381Fr>(ent + 8);
        if (!ok) fail(ZKHIP_ERR_PARSE, "non-canonical field element in the witness");
        std::vector<uint32_t>& v = id >= 0 ? pos : neg;
        const uint64_t k = id >= 0 ? (uint64_t)id : (uint64_t)(-(id + 1));
        if (k >= std::min<uint64_t>((uint64_t)1 << 31, std::max<uint64_t>((uint64_t)1 << 20, 64 * count)))
            fail(ZKHIP_ERR_PARSE, "variable id out of range in the witness");
        if (k >= v.size()) v.resize(std::max<size_t>(k + 1, v.size() * 2), 0);
        if (id < 0 && v[k] == 0) ++n_out;
        v[k] = (uint32_t)(i + 1);
    }
    auto find = [&](int64_t id) -> const uint8_t* {
        const std::vector<uint32_t>& v = id >= 0 ? pos : neg;
        const uint64_t k = id >= 0 ? (uint64_t)id : (uint64_t)(-(id + 1));
        if (k >= v.size() || v[k] == 0) return nullptr;
        return wit + 8 + 40 * (uint64_t)(v[k] - 1) + 8;
    };
    auto name = [](int64_t id) {
        return id == 0 ? std::string("~one") : id > 0 ? "_" + std::to_string(id - 1) : "~out_" + std::to_string(-(id + 1));
    };
    // public_inputs_values: public arguments in argument order, then ~out_0.. by index        ir/mod.rs:278-288, witness.rs:12-23
    // (computed first: the reference evaluates it before proving)
    const uint64_t total = prog->public_args.size() + n_out;
    if (n_inputs) *n_inputs = total;
    if (inputs_out) {
        if (cap < total) fail(ZKHIP_ERR_BAD_ARG, "inputs buffer too small");
        uint64_t k = 0;
        for (int64_t id : prog->public_args) {
            const uint8_t* v = find(id);
            if (!v) fail(ZKHIP_ERR_UNSATISFIED, "assignment missing for public argument " + name(id));
            memcpy(inputs_out + 32 * k++, v, 32);
        }
        for (uint64_t i = 0; i < n_out; ++i) {
            const uint8_t* v = find(-(int64_t)i - 1);
            if (!v) fail(ZKHIP_ERR_UNSATISFIED, "assignment missing for " + name(-(int64_t)i - 1));
            memcpy(inputs_out + 32 * k++, v, 32);
        }
    }
    // z in ark order; every variable is `remove`d from the witness exactly once (AssignmentMissing otherwise)
    if (z_out) {
        const uint64_t m = prog->l + prog->w;
        memset(z_out, 0, 32);
        z_out[0] = 1;
        std::vector<uint8_t> used(count, 0);
        for (uint64_t j = 1; j < m; ++j) {
            const int64_t id = prog->order[j];
            const uint8_t* v = find(id);
            if (!v) fail(ZKHIP_ERR_UNSATISFIED, "assignment missing for variable " + name(id));
            const uint64_t ent = (uint64_t)(v - (wit + 16)) / 40;
            if (used[ent]) fail(ZKHIP_ERR_UNSATISFIED, "assignment for variable " + name(id) + " consumed twice (repeated parameter)");
            used[ent] = 1;
            memcpy(z_out + 32 * j, v, 32);
        }
    }
}

}  // namespace zk
