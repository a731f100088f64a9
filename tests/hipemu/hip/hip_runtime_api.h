// TEST INFRASTRUCTURE: host API part of the HIP stand-in (see hip_runtime.h), enough for examples/cabi_coarse.c
// to be compiled as C++ against the emulated library and run without a GPU.
#pragma once
#include "hip_runtime.h"

static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t *s) { *s = nullptr; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
