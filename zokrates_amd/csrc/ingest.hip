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

#include <chrono>
#include <climits>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>

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
    // text key (definite length only, which is all serde_cbor writes for identifiers): a view into the file
    struct Key {
        const uint8_t* p;
        uint64_t n;
        bool is(const char* s) const { const size_t l = strlen(s); return l == n && !memcmp(p, s, l); }
    };
    Key key(const char* what) {
        const Head h = head();
        if (h.major != 3 || h.indef) fail(ZKHIP_ERR_PARSE, std::string("text string expected for ") + what);
        return Key{take(h.arg), h.arg};
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
    uint32_t find(int64_t id) const {   // after the merge: every id of the program is bound
        const std::vector<uint32_t>& v = id >= 0 ? pos : neg;
        const uint64_t k = id >= 0 ? (uint64_t)id : (uint64_t)(-(id + 1));
        return v[k];
    }
};

// The constraint section is a plain concatenation of CBOR items, and only the ORDER in which variables are first seen
// ties one statement to the next (ark numbers them in that order).  So the section is cut into chunks that are decoded
// independently — each keeps raw ZoKrates ids and the list of ids in the order it first met them — and the chunks' lists,
// walked in file order, reproduce the sequential allocation exactly.  CBOR is not self-synchronising: a chunk starts at
// the first occurrence of the bytes every constraint statement begins with ({"Constraint": = a1 6a "Constraint") after its
// nominal offset, and the cut is only trusted if the chunk before it ends EXACTLY there; any mismatch or error (the
// pattern inside a field element or a directive's payload) sends the whole section through one chunk instead.
template <class P>
struct Builder {
    typedef Fe<P> Fr;
    struct Term {
        int64_t id;
        Fr coeff;
    };
    struct Seen {                       // set of ids, bitmap grown on demand
        std::vector<uint64_t> pos, neg;
        bool test_and_set(int64_t id, uint64_t id_limit) {
            std::vector<uint64_t>& v = id >= 0 ? pos : neg;
            const uint64_t k = id >= 0 ? (uint64_t)id : (uint64_t)(-(id + 1));
            if (k >= id_limit) fail(ZKHIP_ERR_PARSE, "variable id out of range");
            if ((k >> 6) >= v.size()) v.resize(std::max<size_t>((k >> 6) + 1, v.size() * 2), 0);
            const uint64_t bit = (uint64_t)1 << (k & 63);
            const bool was = v[k >> 6] & bit;
            v[k >> 6] |= bit;
            return was;
        }
    };
    struct Chunk {
        const uint8_t *begin = nullptr, *limit = nullptr, *end_reached = nullptr;
        std::vector<Term> terms[3];         // all rows' terms: duplicates merged, zero coefficients dropped
        std::vector<uint32_t> count[3];     // terms per row
        std::vector<int64_t> first_seen;    // ids in the order this chunk first met them
        Seen seen;
        std::vector<Term> scratch;
        std::vector<uint32_t> order;        // lin_comb: the terms of a long row by variable
        bool failed = false;
        IngestError err{0, ""};
        uint64_t id_limit = 0;
        uint64_t row0 = 0, nnz0[3] = {0, 0, 0};   // where this chunk's rows / entries land in the whole program
    };

    static bool canonical(const Fr& x) {
        for (int i = P::N - 1; i >= 0; --i)
            if (x.v[i] != P::mod(i)) return x.v[i] < P::mod(i);
        return false;
    }
    static int64_t variable_id(Cbor& c) {
        int64_t id = 0;
        bool have_id = false;
        uint64_t vf = c.enter(5, "a variable");
        while (c.more(vf)) {
            if (c.key("a Variable field").is("id")) { id = c.integer("Variable.id"); have_id = true; }
            else c.skip();
        }
        if (!have_id) fail(ZKHIP_ERR_PARSE, "Variable without id");
        return id;
    }
    // LinComb {span, value: [[{id}, bytes], ...]} -> one matrix row: duplicates summed (ark's `acc + (coeff, var)`
    // merges equal variables), zero coefficients dropped (ark drops them when the matrices are extracted)
    static void lin_comb(Cbor& c, Chunk& ch, int which) {
        std::vector<Term>& scratch = ch.scratch;
        scratch.clear();
        uint64_t nf = c.enter(5, "a linear combination");
        bool have_value = false;
        while (c.more(nf)) {
            if (!c.key("a LinComb field").is("value")) { c.skip(); continue; }
            have_value = true;
            uint64_t nt = c.enter(4, "LinComb.value");
            while (c.more(nt)) {
                uint64_t pair = c.enter(4, "a (variable, coefficient) pair");
                if (pair != 2) fail(ZKHIP_ERR_PARSE, "LinComb term is not a pair");
                Term t;
                t.id = variable_id(c);
                const Cbor::Head h = c.head();
                if (h.major != 2 || h.indef || h.arg != 32) fail(ZKHIP_ERR_PARSE, "field element is not a 32-byte string");
                memcpy(t.coeff.v, c.take(32), 32);
                if (!canonical(t.coeff)) fail(ZKHIP_ERR_PARSE, "non-canonical field element in the program");
                if (!ch.seen.test_and_set(t.id, ch.id_limit)) ch.first_seen.push_back(t.id);
                scratch.push_back(t);
            }
        }
        if (!have_value) fail(ZKHIP_ERR_PARSE, "LinComb without value");
        // merge duplicates into their first occurrence.  Most rows are tiny and a quadratic pass beats sorting; the reference's
        // optimizer also writes rows of thousands of terms (every linear definition inlined: the sum checks of a SHA-256 round,
        // zokrates_core/src/optimizer/redefinition.rs:24-40), where the quadratic pass was 3/4 of reading such a program
        uint32_t kept = 0;
        constexpr int64_t GONE = INT64_MIN;
        const size_t nterms = scratch.size();
        if (nterms > 24) {
            std::vector<uint32_t>& order = ch.order;
            order.resize(nterms);
            for (size_t i = 0; i < nterms; ++i) order[i] = (uint32_t)i;
            std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return scratch[a].id != scratch[b].id ? scratch[a].id < scratch[b].id : a < b; });
            for (size_t i = 0; i < nterms;) {
                size_t j = i + 1;
                while (j < nterms && scratch[order[j]].id == scratch[order[i]].id) {
                    scratch[order[i]].coeff = fe_add(scratch[order[i]].coeff, scratch[order[j]].coeff);
                    scratch[order[j]].id = GONE;
                    ++j;
                }
                i = j;
            }
        }
        for (size_t i = 0; i < nterms; ++i) {
            if (scratch[i].id == GONE) continue;
            if (nterms <= 24)
                for (size_t j = i + 1; j < nterms; ++j)
                    if (scratch[j].id == scratch[i].id) {
                        scratch[i].coeff = fe_add(scratch[i].coeff, scratch[j].coeff);
                        scratch[j].id = GONE;
                    }
            if (!scratch[i].coeff.is_zero()) { ch.terms[which].push_back(scratch[i]); ++kept; }
        }
        ch.count[which].push_back(kept);
    }
    static void constraint(Cbor& c, Chunk& ch) {
        uint64_t nf = c.enter(5, "a constraint statement");
        int seen = 0;
        // serde writes the fields in declaration order (span, quad, lin, error): quad before lin, which is the order
        // ark_combination is called in (left, right, lin) and hence the order variables are allocated in
        while (c.more(nf)) {
            const Cbor::Key key = c.key("a ConstraintStatement field");
            if (key.is("quad")) {
                if (seen != 0) fail(ZKHIP_ERR_PARSE, "ConstraintStatement fields out of order");
                uint64_t qf = c.enter(5, "a quadratic combination");
                int qseen = 0;
                while (c.more(qf)) {
                    const Cbor::Key qk = c.key("a QuadComb field");
                    if (qk.is("left")) { if (qseen != 0) fail(ZKHIP_ERR_PARSE, "QuadComb fields out of order"); lin_comb(c, ch, 0); qseen = 1; }
                    else if (qk.is("right")) { if (qseen != 1) fail(ZKHIP_ERR_PARSE, "QuadComb fields out of order"); lin_comb(c, ch, 1); qseen = 2; }
                    else c.skip();
                }
                if (qseen != 2) fail(ZKHIP_ERR_PARSE, "QuadComb without left/right");
                seen = 1;
            } else if (key.is("lin")) {
                if (seen != 1) fail(ZKHIP_ERR_PARSE, "ConstraintStatement fields out of order");
                lin_comb(c, ch, 2);
                seen = 2;
            } else {
                c.skip();
            }
        }
        if (seen != 2) fail(ZKHIP_ERR_PARSE, "ConstraintStatement without quad/lin");
    }
    // statements from ch.begin until the reader stands at or beyond ch.limit: a stream of CBOR items, constraints only   lib.rs:115-123
    static void parse_chunk(Chunk& ch, const uint8_t* section_end) {
        try {
            Cbor c{ch.begin, section_end};
            while (c.p < ch.limit) {
                const uint8_t b = c.peek();
                if ((b >> 5) == 3) { c.skip(); continue; }          // a unit variant would be a bare string: none of ours
                uint64_t one = c.enter(5, "a statement");
                if (one != 1) fail(ZKHIP_ERR_PARSE, "statement is not a single-entry map");
                if (c.key("the statement variant").is("Constraint")) constraint(c, ch);
                else c.skip();                                      // Directive, Log: witness generation only
            }
            ch.end_reached = c.p;
        } catch (const IngestError& e) {
            ch.failed = true;
            ch.err = e;
        } catch (const std::bad_alloc&) {
            ch.failed = true;
            ch.err = IngestError{ZKHIP_ERR_NOMEM, "out of host memory"};
        }
    }
    static unsigned worker_count(uint64_t st_len) {
        unsigned hw = std::thread::hardware_concurrency();
        if (const char* e = getenv("ZKHIP_INGEST_THREADS")) hw = (unsigned)std::max(1, atoi(e));
        uint64_t min_chunk = (uint64_t)2 << 20;                     // at least 2 MiB of statements per worker
        if (const char* e = getenv("ZKHIP_INGEST_MIN_CHUNK")) min_chunk = (uint64_t)std::max(16, atoi(e));   // (test hook)
        const uint64_t by_size = st_len / min_chunk;
        return (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(std::min<unsigned>(hw ? hw : 1, 64), by_size));
    }
    // chunk starts: the first {"Constraint": at or after t * len / T
    static std::vector<const uint8_t*> cut_points(const uint8_t* b, const uint8_t* e, unsigned T) {
        static const uint8_t PAT[12] = {0xa1, 0x6a, 'C', 'o', 'n', 's', 't', 'r', 'a', 'i', 'n', 't'};
        std::vector<const uint8_t*> cuts{b};
        const uint64_t len = (uint64_t)(e - b);
        for (unsigned t = 1; t < T; ++t) {
            const uint8_t* from = std::max(cuts.back() + 1, b + len * t / T);
            const uint8_t* hit = nullptr;
            for (const uint8_t* q = from; q + sizeof(PAT) <= e; ++q) {
                q = (const uint8_t*)memchr(q, 0xa1, (size_t)(e - q) - sizeof(PAT) + 1);
                if (!q) break;
                if (!memcmp(q, PAT, sizeof(PAT))) { hit = q; break; }
            }
            if (!hit) break;
            cuts.push_back(hit);
        }
        cuts.push_back(e);
        return cuts;
    }

    void run(const uint8_t* bytes, uint64_t par_off, uint64_t par_len, uint64_t st_off, uint64_t st_len, zkhip_prog* out) {
        Symbols sym;
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
                    const Cbor::Key key = c.key("a Parameter field");
                    if (key.is("id")) {
                        id = variable_id(c);
                        have_id = true;
                    } else if (key.is("private")) {
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
        // statements: decoded in chunks, in parallel when the section is large
        const uint8_t *sb = bytes + st_off, *se = sb + st_len;
        std::vector<Chunk> chunks;
        auto decode = [&](unsigned T) {
            const std::vector<const uint8_t*> cuts = cut_points(sb, se, T);
            chunks.clear();
            chunks.resize(cuts.size() - 1);
            for (size_t t = 0; t + 1 < cuts.size(); ++t) {
                chunks[t].begin = cuts[t];
                chunks[t].limit = cuts[t + 1];
                chunks[t].id_limit = sym.id_limit;
            }
            if (chunks.size() == 1) {
                parse_chunk(chunks[0], se);
            } else {
                HostThreads th;
                for (size_t t = 1; t < chunks.size(); ++t) th.run([&, t] { parse_chunk(chunks[t], se); });
                parse_chunk(chunks[0], se);
                th.join();
            }
            for (size_t t = 0; t < chunks.size(); ++t)
                if (chunks[t].failed || chunks[t].end_reached != chunks[t].limit) return false;
            return true;
        };
        const unsigned T = worker_count(st_len);
        const bool prof = getenv("ZKHIP_INGEST_PROFILE") != nullptr;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t_begin = now();
        const bool cut_ok = decode(T);
        const auto t_decoded = now();
        if (prof) fprintf(stderr, "[zkhip ingest] %u worker(s) asked, %zu chunk(s), cuts %s, decode %.1f ms\n", T, chunks.size(), cut_ok ? "ok" : "REJECTED", ms(t_begin, t_decoded));
        if (!cut_ok) {
            if (chunks.size() > 1) decode(1);            // a cut inside an item (or a real error): one chunk decides
            if (chunks[0].failed) throw chunks[0].err;
            if (chunks[0].end_reached != se) fail(ZKHIP_ERR_PARSE, "program file truncated inside a CBOR item");
        }
        // the sequential part: variables in first-seen order over the chunks in file order
        for (Chunk& ch : chunks) {
            for (int64_t id : ch.first_seen) sym.lookup(id);
            std::vector<int64_t>().swap(ch.first_seen);
            Seen().pos.swap(ch.seen.pos);
            Seen().neg.swap(ch.seen.neg);
        }
        const auto t_merged = now();
        const uint64_t l = sym.inst.size(), w = sym.wit.size();
        if (l + w + 2 >= ((uint64_t)1 << 31)) fail(ZKHIP_ERR_BAD_ARG, "too many variables");
        uint64_t n = 0, nnz[3] = {0, 0, 0};
        for (Chunk& ch : chunks) {
            ch.row0 = n;
            n += ch.count[0].size();
            for (int k = 0; k < 3; ++k) { ch.nnz0[k] = nnz[k]; nnz[k] += ch.terms[k].size(); }
        }
        out->n = n; out->l = l; out->w = w;
        out->order = sym.inst;
        out->order.insert(out->order.end(), sym.wit.begin(), sym.wit.end());
        for (int k = 0; k < 3; ++k) {
            out->rp[k].resize(n + 1);
            out->rp[k][n] = nnz[k];
            out->col[k].resize(nnz[k]);
            out->val[k].resize(nnz[k] * 32);
        }
        const auto t_alloc = now();
        if (prof) fprintf(stderr, "[zkhip ingest] output arrays %.1f ms\n", ms(t_merged, t_alloc));
        // columns, values and row pointers of every chunk's rows (independent: in parallel)
        auto emit = [&](Chunk& ch) {
            for (int k = 0; k < 3; ++k) {
                uint64_t q = ch.nnz0[k];
                const Term* t = ch.terms[k].data();
                uint32_t* col = out->col[k].data();
                uint8_t* val = out->val[k].data();
                std::vector<std::pair<uint32_t, uint32_t>> byc;      // (column, position in the row) of a long row
                for (size_t r = 0; r < ch.count[k].size(); ++r) {
                    out->rp[k][ch.row0 + r] = q;
                    const uint64_t a = q;
                    const uint32_t cnt = ch.count[k][r];
                    if (cnt > 24) {
                        // a long row (compiled programs have combinations of thousands of terms, in whatever order the compiler
                        // left them): the insertion sort below would move 32-byte values O(k^2) times — sort (column, position)
                        // pairs and place every value once
                        byc.resize(cnt);
                        for (uint32_t j = 0; j < cnt; ++j) {
                            const uint32_t tag = sym.find(t[j].id);
                            byc[j] = {(tag & WIT_TAG) ? (uint32_t)(l + (tag & ~WIT_TAG)) : tag, j};
                        }
                        std::sort(byc.begin(), byc.end());
                        for (uint32_t j = 0; j < cnt; ++j) {
                            col[q + j] = byc[j].first;
                            memcpy(val + (q + j) * 32, t[byc[j].second].coeff.v, 32);
                        }
                        t += cnt;
                        q += cnt;
                        continue;
                    }
                    for (uint32_t j = 0; j < cnt; ++j, ++t, ++q) {
                        const uint32_t tag = sym.find(t->id);
                        const uint32_t cj = (tag & WIT_TAG) ? (uint32_t)(l + (tag & ~WIT_TAG)) : tag;
                        // ark keeps a row sorted by variable (One, Instance(i), Witness(i)): same order as the final column
                        // index — insertion sort, rows are tiny
                        uint64_t y = q;
                        while (y > a && col[y - 1] > cj) {
                            col[y] = col[y - 1];
                            memcpy(val + y * 32, val + (y - 1) * 32, 32);
                            --y;
                        }
                        col[y] = cj;
                        memcpy(val + y * 32, t->coeff.v, 32);
                    }
                }
                std::vector<Term>().swap(ch.terms[k]);
                std::vector<uint32_t>().swap(ch.count[k]);
            }
        };
        if (chunks.size() == 1) {
            emit(chunks[0]);
        } else {
            HostThreads th;
            for (size_t t = 1; t < chunks.size(); ++t) th.run([&, t] { emit(chunks[t]); });
            emit(chunks[0]);
            th.join();
        }
        if (prof) fprintf(stderr, "[zkhip ingest] first-seen merge %.1f ms, emit %.1f ms\n", ms(t_decoded, t_merged), ms(t_merged, now()));
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

// ------------------------------------------------------------------ the inverse: an R1CS as a ZoKrates `out` program
// `ProgIterator::serialize` (/root/reference/zokrates_ast/src/ir/serialize.rs:202-279) for a program whose statements are
// exactly the constraints A_i z * B_i z = C_i z in row order, no directives, no solvers: what `zokrates` tooling needs to
// consume a circuit that arrived as matrices (an iden3 .r1cs file, a synthetic benchmark circuit).  ids[j] names column j
// (0 = ~one, k > 0 = _{k-1}, -k = ~out_{k-1}); a reader allocates variables in first-seen order (zokrates_ark/src/lib.rs:
// 80-129), so the columns come back in the caller's order only if that is the order in which the rows mention them.
namespace {
struct CborOut {
    uint8_t* p;
    uint8_t* e;
    void need(size_t n) { if ((size_t)(e - p) < n) fail(ZKHIP_ERR_BAD_ARG, "output buffer too small (see zkhip_prog_write_bound)"); }
    void raw(const void* b, size_t n) { need(n); memcpy(p, b, n); p += n; }
    void byte(uint8_t b) { need(1); *p++ = b; }
    void head(int major, uint64_t v) {
        if (v < 24) { byte((uint8_t)(major << 5 | v)); return; }
        const int nb = v < (1u << 8) ? 1 : v < (1u << 16) ? 2 : v < ((uint64_t)1 << 32) ? 4 : 8;
        byte((uint8_t)(major << 5 | (nb == 1 ? 24 : nb == 2 ? 25 : nb == 4 ? 26 : 27)));
        for (int i = nb - 1; i >= 0; --i) byte((uint8_t)(v >> (8 * i)));
    }
    void text(const char* t) { const size_t n = strlen(t); head(3, n); raw(t, n); }
    void null() { byte(0xf6); }
    void integer(int64_t v) { if (v >= 0) head(0, (uint64_t)v); else head(1, (uint64_t)(-(v + 1))); }
    void variable(int64_t id) { head(5, 1); text("id"); integer(id); }
};
}  // namespace

uint64_t prog_write_bound(uint64_t n, uint64_t nnz, uint64_t n_args) {
    // header region + per argument + per statement (keys, nulls, container heads) + per term (pair, variable, 32-byte string)
    return 120 + 16 + n_args * 48 + n * 96 + nnz * 56 + 64;
}

uint64_t prog_write(int curve, uint64_t n, uint64_t m, const uint64_t* const rp[3], const uint32_t* const col[3], const uint8_t* const val[3],
                    const int64_t* ids, const int64_t* arg_ids, const uint8_t* arg_private, uint64_t n_args, uint32_t return_count, uint8_t* out,
                    uint64_t cap) {
    constexpr uint64_t HEADER_REGION = 120;   // size_of::<ProgHeader>() on x86-64; the 100 bytes the header occupies lead it
    if (curve != ZKHIP_CURVE_BN128 && curve != ZKHIP_CURVE_BLS12_381) fail(ZKHIP_ERR_BAD_ARG, "unknown curve id");
    if (n >= ((uint64_t)1 << 32)) fail(ZKHIP_ERR_BAD_ARG, "constraint count does not fit the header");
    if (cap < HEADER_REGION) fail(ZKHIP_ERR_BAD_ARG, "output buffer too small (see zkhip_prog_write_bound)");
    memset(out, 0, HEADER_REGION);
    CborOut c{out + HEADER_REGION, out + cap};
    uint64_t off[4], len[4];
    // parameters: [{span, id: {id}, private}]
    off[0] = (uint64_t)(c.p - out);
    c.head(4, n_args);
    for (uint64_t a = 0; a < n_args; ++a) {
        c.head(5, 3);
        c.text("span"); c.null();
        c.text("id"); c.variable(arg_ids[a]);
        c.text("private"); c.byte(arg_private[a] ? 0xf5 : 0xf4);
    }
    len[0] = (uint64_t)(c.p - out) - off[0];
    // constraints: a stream of {"Constraint": {span, quad: {span, left, right}, lin, error}}
    off[1] = (uint64_t)(c.p - out);
    auto lin_comb = [&](int k, uint64_t i) {
        c.head(5, 2);
        c.text("span"); c.null();
        c.text("value");
        const uint64_t a = rp[k][i], b = rp[k][i + 1];
        if (b < a) fail(ZKHIP_ERR_BAD_ARG, "rowptr not monotone");
        c.head(4, b - a);
        for (uint64_t q = a; q < b; ++q) {
            if (col[k][q] >= m) fail(ZKHIP_ERR_BAD_ARG, "column index out of range");
            c.head(4, 2);
            c.variable(ids[col[k][q]]);
            c.head(2, 32);
            c.raw(val[k] + q * 32, 32);
        }
    };
    for (uint64_t i = 0; i < n; ++i) {
        c.head(5, 1);
        c.text("Constraint");
        c.head(5, 4);
        c.text("span"); c.null();
        c.text("quad");
        c.head(5, 3);
        c.text("span"); c.null();
        c.text("left"); lin_comb(0, i);
        c.text("right"); lin_comb(1, i);
        c.text("lin"); lin_comb(2, i);
        c.text("error"); c.null();
    }
    len[1] = (uint64_t)(c.p - out) - off[1];
    off[2] = (uint64_t)(c.p - out);
    c.head(4, 0);                                   // solvers: []
    len[2] = 1;
    off[3] = (uint64_t)(c.p - out);
    c.head(5, 1); c.text("modules"); c.head(5, 0);  // module map: {modules: {}}
    len[3] = (uint64_t)(c.p - out) - off[3];
    // the header (ProgHeader::write, serialize.rs:133-148); the module map is tagged as a second Solvers section (:247)
    uint8_t* h = out;
    static const uint8_t MAGIC_VERSION[8] = {0x5a, 0x4f, 0x4b, 0, 3, 0, 0, 0};
    memcpy(h, MAGIC_VERSION, 8);
    memcpy(h + 8, curve == ZKHIP_CURVE_BN128 ? ID_BN128 : ID_BLS12_381, 4);
    const uint32_t cnt = (uint32_t)n;
    memcpy(h + 12, &cnt, 4);
    memcpy(h + 16, &return_count, 4);
    static const uint32_t TYPES[4] = {1, 2, 3, 3};
    for (int s = 0; s < 4; ++s) {
        memcpy(h + 20 + 20 * s, &TYPES[s], 4);
        memcpy(h + 24 + 20 * s, &off[s], 8);
        memcpy(h + 32 + 20 * s, &len[s], 8);
    }
    return (uint64_t)(c.p - out);
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
