// Render kernels of the Pathtrace integrator (every layout, variant and both kernel families; with and without the statistics counters).
#include "hjr_launch.hip.h"
template int hjr_launch<HJR_INTEGRATOR_PT, false>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
template int hjr_launch<HJR_INTEGRATOR_PT, true>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
