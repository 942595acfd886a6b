// stand-in for <cuda.h> when the reference kernels are compiled for the CPU (oracle/_ref)
#pragma once
#include "cuda_cpu_shim.h"
