// step_diag.hip -- the step kernels with the afc::diag instantiation of the optimised phases (see step_kernels.h)
#define AFC_MODE 1
#define AFC_MODE_NS diag
#include "step_kernels.h"
