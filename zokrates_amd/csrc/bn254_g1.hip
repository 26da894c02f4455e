// bn254_g1.hip — the G1 kernels of BN254 (bucket accumulation, fold, fixed-base) in a translation unit of their own.
#include "group.cuh"
namespace zk {
ZK_INSTANTIATE_GROUP(Fe<Bn254Fq>)
ZK_INSTANTIATE_BIND(Fe<Bn254Fq>)
}  // namespace zk
