/* smpl_amd/csrc/test_hooks.h -- entry points that exist for the parity tests only.  They are exported by the shared
 * library but are NOT part of the drop-in boundary (include/smpl_amd.h): nothing a planner needs is declared here. */
#pragma once

#include "../../include/smpl_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Shrinks the (edge, waypoint) work list of the expansion pipeline to `items` entries, so that nearly every edge
 * overflows it and is walked whole by its finish thread (the deferred pass, which an ordinary batch never needs);
 * 0 restores the default size. */
int smplx_test_set_work_list_items(smplx_space* s, int items);

#ifdef __cplusplus
}
#endif
