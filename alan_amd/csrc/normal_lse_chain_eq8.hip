// The chained launch's kernels for event-length bucket EQ = 8 (normal_lse_chain_impl.h).
#include "normal_lse_chain_impl.h"

namespace alan {
int chain_launch_eq8(const ChainPlan &p, hipStream_t stream) { return chain_launch_eq<8>(p, stream); }
}  // namespace alan
