// The traversal kernels with (record, side) stack entries (namespace ptd_rs): pt_kernels.hip compiled once more with
// PT_STACK_CHILD 0, for scenes in global memory whose child references do not fit the packed (ref << 7 | cnt) entry
// (a leaf of 128 triangles or more, or 2^25 records / triangles), and for PT_CHILD_ENTRIES=0. Same arithmetic, same
// results; a pop costs one more 8-byte fetch. Why this is a compilation of its own and not a run-time switch: see the
// note on PT_STACK_CHILD in pt_kernels.hip.
#define PT_STACK_CHILD 0
#include "pt_kernels.hip"
