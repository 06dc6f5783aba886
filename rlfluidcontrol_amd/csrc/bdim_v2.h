// bdim_v2.h -- the optimised phases of the per-env BDIM step (bdim_v2_body.h), compiled twice:
//   afc::prod  the shipped instantiation: no experiment switches, no phase marks, no stop codes, no gates
//   afc::diag  the same code with the diagnostics of tools/ compiled in (StepParams.dbg, Prof marks, stop codes,
//              streaming gates); selected at launch time when cfg.reserved[0] != 0, a start stagger is configured or
//              phase profiling is enabled.
#pragma once
#include "bdim_device.h"

#define AFC_DIAG 0
#define AFC_NS prod
#include "bdim_v2_body.h"
#undef AFC_DIAG
#undef AFC_NS

#define AFC_DIAG 1
#define AFC_NS diag
#include "bdim_v2_body.h"
#undef AFC_DIAG
#undef AFC_NS
