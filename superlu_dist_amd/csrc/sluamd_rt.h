// HIP runtime for the product build (the CPU test build under oracle/emul/ shadows this header with a host shim)
#pragma once
#include <hip/hip_runtime.h>
