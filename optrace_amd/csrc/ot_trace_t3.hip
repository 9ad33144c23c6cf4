// trace_tail_kernel variants (render-only chunks) of feature level OT_FEAT(OT_HIT_ILLINOIS, 1): aspheres and tilted surfaces + HURB
#include "ot_trace_kernel.hpp"

OT_DEFINE_TRACE_TAIL_LAUNCHER(OT_FEAT(OT_HIT_ILLINOIS, 1))
