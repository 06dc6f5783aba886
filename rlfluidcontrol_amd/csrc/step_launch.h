// step_launch.h -- host entry points of the four step-kernel translation units (step_kernels.h)
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "bdim_device.h"

namespace afc {
#define AFC_DECL_MODE(ns)                                                                                              \
  void launch_sim_step_##ns(int cpl, int E, size_t lds, hipStream_t st, const StepParams *dP, const float *dxi,        \
                            float *dforce);                                                                            \
  void launch_agent_step_##ns(int cpl, int E, size_t lds, hipStream_t st, const StepParams *dP, const float *dact,     \
                              float *dobs, float *drew, uint8_t *ddone, int *dns, int max_steps);
AFC_DECL_MODE(prod)
AFC_DECL_MODE(diag)
AFC_DECL_MODE(perf)
AFC_DECL_MODE(lat)
#undef AFC_DECL_MODE
void launch_op_advdif_prod(size_t lds, hipStream_t st, const StepParams &P, int same);
void launch_op_advdif_diag(size_t lds, hipStream_t st, const StepParams &P, int same);
void launch_op_project_prod(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl, float *dshift);
void launch_op_project_diag(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl, float *dshift);
void launch_op_project_perf(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl, float *dshift);
void launch_op_mg_perf(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl);
void launch_op_mg_prod(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl);
void launch_op_mg_diag(size_t lds, hipStream_t st, const StepParams &P, int *dit, float *dl);
int occupancy_agent_step_diag(int cpl, size_t lds, int *per_cu);
}  // namespace afc
