// rows kernel: LDS-staged fast path (see reduce.hip header).  Placeholder until implemented:
// every problem is routed to the group kernel.
#include "plan.h"

namespace alan {

int try_launch_rows(const Canon &, int, int, int, double, hipStream_t) { return ALAN_ERR_UNSUPPORTED; }

}  // namespace alan
