/* od_common.cuh - shared host-side helpers of libdaalahip (error mapping). */
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../../include/daala_hip.h"

/* Maps the sticky HIP launch status to the reference's error convention
   (include/daala/codec.h:89-103).  No CPU fallback exists: a failure is
   reported, never papered over. */
static inline int odhip_check_launch(void) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "libdaalahip: kernel launch failed: %s\n", hipGetErrorString(e));
    return ODHIP_EFAULT;
  }
  return ODHIP_SUCCESS;
}

#define ODHIP_TRY(expr) \
  do { \
    hipError_t e_ = (expr); \
    if (e_ != hipSuccess) { \
      fprintf(stderr, "libdaalahip: %s failed: %s\n", #expr, hipGetErrorString(e_)); \
      return ODHIP_EFAULT; \
    } \
  } while (0)
