// stand-in for <ATen/cuda/CUDAContext.h> when the reference kernels are compiled for the CPU (oracle/_ref)
#pragma once
#include "cuda_cpu_shim.h"
