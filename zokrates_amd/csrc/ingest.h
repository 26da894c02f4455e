// ingest.h — host-side readers of ZoKrates' own input files ("next" row N1 of SURVEY.md §8f): the compiled program
// `out` and the `witness` file, turned into the R1CS and the assignment in ark variable order.  Pure host code (no
// device work); the C ABI wrappers are in zkhip_api.hip.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

// a vector whose resize() leaves new elements uninitialised: the matrices are hundreds of megabytes that the reader's workers
// write in full, and value-initialising them first was one serial pass over all of it (100 ms of a 2^20-constraint program
// whatever the thread count: the part of the reader that did not scale)
template <class T>
struct zkhip_default_init : std::allocator<T> {
    template <class U> struct rebind { typedef zkhip_default_init<U> other; };
    template <class U> void construct(U* p) { ::new ((void*)p) U; }
    template <class U, class A0, class... A> void construct(U* p, A0&& a0, A&&... a) { ::new ((void*)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};
template <class T> using zkhip_raw_vector = std::vector<T, zkhip_default_init<T>>;

struct zkhip_prog {
    int curve = 0;
    uint64_t n = 0, l = 0, w = 0, return_count = 0;
    zkhip_raw_vector<uint64_t> rp[3];
    zkhip_raw_vector<uint32_t> col[3];
    zkhip_raw_vector<uint8_t> val[3];      // canonical LE, 32 B per entry
    std::vector<int64_t> order;            // ZoKrates variable id (flat/variable.rs: 0 = ~one, k > 0 = _{k-1}, -k = ~out_{k-1}) of column j
    std::vector<int64_t> public_args;      // ids of the public arguments, in argument order
};

namespace zk {
struct IngestError {
    int32_t code;
    std::string msg;
};
// ProgEnum::deserialize + Computation::generate_constraints; throws IngestError
void prog_parse(const uint8_t* bytes, size_t len, zkhip_prog* out);
// Witness::read + the `witness.remove(..)` walk of generate_constraints (z, m x 32 B) + public_inputs_values
// (inputs_out, up to cap elements; *n_inputs = how many there are); throws IngestError
void prog_assignment(const zkhip_prog* prog, const uint8_t* wit, size_t len, uint8_t* z_out, uint8_t* inputs_out, uint64_t cap, uint64_t* n_inputs);
// R1CS -> the bytes of a ZoKrates `out` program (ProgIterator::serialize); returns the length written; throws IngestError
uint64_t prog_write_bound(uint64_t n, uint64_t nnz, uint64_t n_args);
uint64_t prog_write(int curve, uint64_t n, uint64_t m, const uint64_t* const rp[3], const uint32_t* const col[3], const uint8_t* const val[3],
                    const int64_t* ids, const int64_t* arg_ids, const uint8_t* arg_private, uint64_t n_args, uint32_t return_count, uint8_t* out,
                    uint64_t cap);
}  // namespace zk
