// step_perf.hip -- the step kernels with the afc::perf instantiation of the optimised phases (see step_kernels.h)
#define AFC_MODE 2
#define AFC_MODE_NS perf
#include "step_kernels.h"
