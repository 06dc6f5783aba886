// bdim_v2.h -- the optimised phases of the per-env BDIM step (bdim_v2_body.h), compiled four times:
//   afc::prod  the shipped instantiation: no experiment switches, no phase marks, no stop codes, no gates
//   afc::perf  the shipped code with the red-black smoother (perf mode, see below)
//   afc::diag  the same code with the diagnostics of tools/ compiled in (StepParams.dbg, Prof marks, stop codes,
//              streaming gates); selected at launch time when cfg.reserved[0] != 0, a start stagger is configured or
//              phase profiling is enabled.
#pragma once
#include "bdim_device.h"

// AFC_PRIO: rotating base priorities of the workgroups that share a CU (prio_rotate in bdim_v2_body.h), every 2^AFC_PRIO
// sim-steps of a launch; -1 = all at 0 (round 5).  Measured on the same memory, 1024 envs of 256x128, per launch:
// 0: +0.5 .. -0.1 %, 1: -0.7 .. -1.5 % (eight boxes), 2: -0.2 .. -0.4 %; 128x64: -1.2 %.  afc::lat and afc::perf keep -1
// (launches of few envs per CU have nothing to balance, and afc::lat's code answers any edit with +-0.5 %).
#ifndef AFC_PRIO_ROT
#define AFC_PRIO_ROT 1
#endif

// AFC_ONLY_MODE (0 / 1 / 2) restricts the header to one instantiation (the step-kernel translation units)
#if !defined(AFC_ONLY_MODE) || AFC_ONLY_MODE == 0
#define AFC_DIAG 0
#define AFC_RB 0
#define AFC_TR 0
#define AFC_PRIO AFC_PRIO_ROT
#define AFC_NS prod
#include "bdim_v2_body.h"
#undef AFC_DIAG
#undef AFC_RB
#undef AFC_TR
#undef AFC_PRIO
#undef AFC_NS
#endif

// afc::lat: the shipped code with the up-legs of the multigrid levels marching DOWN THE ROWS (mg_up_tr, round 6) instead of
// round 5's strip pipeline: the shorter latency chain (one env alone -14 % per launch, one env per CU -9 %), which four
// workgroups per CU do not need (+3 % there).  The host picks it for launches of fewer than 2.5 envs per CU
// (afc_config.upleg).  An instantiation of its own: compiled into afc::prod as well, its call sites alone cost the
// 1024-env launch 1.9 % (the register allocation of the one big step kernel).
#if !defined(AFC_ONLY_MODE) || AFC_ONLY_MODE == 3
#define AFC_DIAG 0
#define AFC_RB 0
#define AFC_TR 1
#define AFC_PRIO (-1)
#define AFC_NS lat
#include "bdim_v2_body.h"
#undef AFC_DIAG
#undef AFC_RB
#undef AFC_TR
#undef AFC_PRIO
#undef AFC_NS
#endif

#if !defined(AFC_ONLY_MODE) || AFC_ONLY_MODE == 1
#define AFC_DIAG 1
#define AFC_RB 0
#define AFC_TR 1
#define AFC_PRIO AFC_PRIO_ROT
#define AFC_NS diag
#include "bdim_v2_body.h"
#undef AFC_DIAG
#undef AFC_RB
#undef AFC_TR
#undef AFC_PRIO
#undef AFC_NS
#endif

// afc::perf: PERF MODE (cfg.gs_mode = 1): MG.smooth's four Gauss-Seidel sweeps in red-black order by all four waves
// instead of the reference's lexicographic order by one marching wave -- a different (equally convergent) smoother, so
// results agree with the reference to a tolerance, not bit for bit.  Never the headline number.
#if !defined(AFC_ONLY_MODE) || AFC_ONLY_MODE == 2
#define AFC_DIAG 0
#define AFC_RB 1
#define AFC_TR 0
#define AFC_PRIO (-1)
#define AFC_NS perf
#include "bdim_v2_body.h"
#undef AFC_DIAG
#undef AFC_RB
#undef AFC_TR
#undef AFC_PRIO
#undef AFC_NS
#endif
