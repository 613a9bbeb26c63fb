// ops_small.hip — kernels of the small models (double integrators, Cartpole) except their forward-pass variants.
#include "ops.h"

namespace to {
template <class M>
static void fill_one(ModelOps& o) {
  fill_misc<M>(o);
  o.expand = op_expand<M>;
  o.backward = op_backward<M>;
  if constexpr (!M::lie && Coop<M>::R <= 8) o.expand_backward_coop = op_expand_backward_coop<M>;
}
void fill_ops_small(ModelOps* t) {
  fill_one<DoubleIntegratorModel<1>>(t[0]);
  fill_one<DoubleIntegratorModel<2>>(t[1]);
  fill_one<DoubleIntegratorModel<3>>(t[2]);
  fill_one<CartpoleModel>(t[3]);
}
}  // namespace to
