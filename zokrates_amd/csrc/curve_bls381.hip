// curve_bls381.hip — instantiates the prover for BLS12-381 (/root/reference/zokrates_field/src/bls12_381.rs:1-13).
#include "core.cuh"
namespace zk {
const CurveOps* curve_ops_bls381() {
    static const CurveOps ops = make_curve_ops<CurveBls381>();
    return &ops;
}
}  // namespace zk
