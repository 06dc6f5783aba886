// step_prod.hip -- the step kernels with the afc::prod instantiation of the optimised phases (see step_kernels.h)
#define AFC_MODE 0
#define AFC_MODE_NS prod
#include "step_kernels.h"
