// trace_kernel variants of feature level OT_FEAT(OT_HIT_SPLINE, 0) (ot_trace_kernel.hpp)
#include "ot_trace_kernel.hpp"

OT_DEFINE_TRACE_LAUNCHER(OT_FEAT(OT_HIT_SPLINE, 0))
