// jm_lib_constraint.cpp -- second translation unit of the per-topology HIP library: the constraint-model
// kernel (jm_constraint.h), compiled in parallel with jm_lib.cpp (which declares the same instantiation
// `extern` under -DJM_SPLIT_CONSTRAINT) because it is the longest single compile of a large topology.
#include <hip/hip_runtime.h>

#ifndef JM_TOPO_HEADER
#error "JM_TOPO_HEADER must name the generated topology header"
#endif
#include JM_TOPO_HEADER

#include "jm_kernels.h"
#include "jm_constraint.h"

namespace jm
{
template __global__ void k_constrained<double, Topo>(const BatchArgs<double>, const ConArgs<double>);
}
