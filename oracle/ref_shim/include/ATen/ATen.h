// stand-in for <ATen/ATen.h> when the reference kernels are compiled for the CPU (oracle/_ref)
#pragma once
#include "cuda_cpu_shim.h"
