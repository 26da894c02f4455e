// curve_bn254.hip — instantiates the prover for BN254 ("bn128", /root/reference/zokrates_field/src/bn128.rs:1-13).
#include "core.cuh"
namespace zk {
const CurveOps* curve_ops_bn254() {
    static const CurveOps ops = make_curve_ops<CurveBn254>();
    return &ops;
}
}  // namespace zk
