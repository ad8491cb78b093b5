// mjs_push.h — the two instances of the Planar-Push kernel (mjs_push_impl.h): block slots are a compile-time
// constant (static contact slots, unrolled per-block solvers, the cooperative workspace in LDS is sized by them).
//   pp  : 2 slots, n_objects <= 2 — the registered env (robot_planar_push.py:315) and BASELINE config 4
//   pp5 : 5 slots, n_objects 3..5 — the reference's dataclass default (robot_planar_push.py:61)
#pragma once
#include "mjs_kernel_common.h"
#include "mjs_reach.h"

#define MJS_PP_NS pp
#define MJS_PP_NB MJS_PP_FAST_OBJECTS
#include "mjs_push_impl.h"
#undef MJS_PP_NS
#undef MJS_PP_NB

#define MJS_PP_NS pp5
#define MJS_PP_NB MJS_PP_MAX_OBJECTS
#include "mjs_push_impl.h"
#undef MJS_PP_NS
#undef MJS_PP_NB
