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

/* The heap primitives of the device-resident search (search_kernel.h) driven by an op sequence in the language of
 * oracle/heap_ref_driver.cpp (pairs code, key; see k_heap_ops): top_after[i] = element at the top after op i, -1 when empty.
 * lds_entries = how many leading heap entries live in LDS (the rest in HBM), 1 .. 4096. */
int smplx_test_heap_ops(const int32_t* ops, int nops, int lds_entries, int32_t* top_after);

/* First capacity (states) of the device-resident search's buffers, so that a test can make a search outgrow them
 * (SMPLX_SS_GROW: the host enlarges and launches again); 0 restores the default sizing. */
int smplx_test_set_search_capacity(smplx_space* s, int states);

/* on = 0: the device-resident search runs WITHOUT its helper wave (the search wave does the successors' bookkeeping inline,
 * as it does for robots whose block would exceed 512 threads with one); 1 restores the default. */
int smplx_test_set_search_helper(smplx_space* s, int on);

#ifdef __cplusplus
}
#endif
