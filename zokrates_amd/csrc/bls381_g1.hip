// bls381_g1.hip — the G1 kernels of BLS12-381 (bucket accumulation, fold, fixed-base) in a translation unit of their own.
#include "group.cuh"
namespace zk {
ZK_INSTANTIATE_GROUP(Fe<Bls381Fq>)
ZK_INSTANTIATE_BIND(Fe<Bls381Fq>)
}  // namespace zk
