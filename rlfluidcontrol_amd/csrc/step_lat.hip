// step_lat.hip -- the step kernels with the afc::lat instantiation of the optimised phases: the shipped code with the
// multigrid up-legs marching down the rows (mg_up_tr) -- what launches of few envs per CU run (see step_kernels.h, bdim_v2.h)
#define AFC_MODE 3
#define AFC_MODE_NS lat
#include "step_kernels.h"
