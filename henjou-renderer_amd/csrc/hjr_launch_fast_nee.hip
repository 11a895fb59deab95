// Approximate-arithmetic render kernels (HJR_FLAG_FAST_MATH) of one integrator: compiled without -fhip-fp32-correctly-rounded-divide-sqrt, with
// -freciprocal-math -fapprox-func -ffp-contract=fast and HJR_FAST_MATH (hjr_math.hip.h: hardware sine / cosine / power); Makefile: FASTFLAGS.
#ifndef HJR_FAST_MATH
#error "compile this unit with -DHJR_FAST_MATH (Makefile: build/hjr_launch_fast_%.o)"
#endif
#include "hjr_launch.hip.h"
template int hjr_launch_fast<HJR_INTEGRATOR_NEE>(hjr_ctx*, const KParams&, uint64_t, int, hipStream_t);
