// bls381_g2.hip — the G2 kernels of BLS12-381 (bucket accumulation, fold, fixed-base) in a translation unit of their own.
#include "group.cuh"
namespace zk {
ZK_INSTANTIATE_GROUP(Fe2<Bls381Fq>)
}  // namespace zk
