// jm_lib_constraint.cpp -- further translation units of the per-topology HIP library: the constraint-model
// kernels, compiled in parallel with jm_lib.cpp (which declares the same instantiations `extern` under
// -DJM_SPLIT_CONSTRAINT) because they are the longest single compiles of a large topology.
//   -DJM_CON_PART=1  k_constrained (jm_constraint.h, one robot per lane: trees without the 4-limb structure)
//   -DJM_CON_PART=2  k_quad_con<0>  (jm_qcon.h, branch-parallel: ANYmal, Atlas, ...: every mode but start / reset)
//   -DJM_CON_PART=3  k_quad_gen     (jm_quad.h with per-lane body parameters / height-map ground / applied forces)
//   -DJM_CON_PART=4  k_quad_con_gen<0> (the same for the constraint contact model)
//   -DJM_CON_PART=5  k_quad_dopri   (jm_qdopri.h, the persistent adaptive stepper)
//   -DJM_CON_PART=6  k_quad_dopri_gen (the same with per-lane body parameters / height map / applied forces)
//   -DJM_CON_PART=7 / 8  k_quad_con_split<1 / 2>  (jm_qcon.h, split stepping of robots with large solves: before / after the solve)
//   -DJM_CON_PART=9  k_qcon_pgs     (the solve of the split form)
//   -DJM_CON_PART=10 k_qtip_pgs     (the solve of the split form in the operational space of the tip bodies, jm_qtip.h)
//   -DJM_CON_PART=11 / 12  k_quad_con<1> / k_quad_con_gen<1>: the `start` / `reset` launches (Engine::start's four passes) as kernels
//                    of their own, so that their code does not weigh on the register allocation of the step path
//   -DJM_CON_PART=13 / 14  k_quad_con_split<1 / 2, INIT = 1>: the same for the split form (7 / 8 are the step parts, INIT = 0)
#include <hip/hip_runtime.h>

#ifndef JM_TOPO_HEADER
#error "JM_TOPO_HEADER must name the generated topology header"
#endif
#include JM_TOPO_HEADER

#include "jm_kernels.h"
#include "jm_constraint.h"
#include "jm_qcon.h"
#if JM_CON_PART == 5 || JM_CON_PART == 6
#include "jm_qdopri.h"
#endif

namespace jm
{
#if JM_CON_PART == 1
template __global__ void k_constrained<double, Topo, false>(const BatchArgs<double>, const ConArgs<double>);
#elif JM_CON_PART == 2 && JM_TOPO_QUAD
template __global__ void k_quad_con<double, Topo, 0>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 3 && JM_TOPO_QUAD
template __global__ void k_quad_gen<double, Topo>(const BatchArgs<double>);
#elif JM_CON_PART == 4 && JM_TOPO_QUAD
template __global__ void k_quad_con_gen<double, Topo, 0>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 5 && JM_TOPO_QUAD
template __global__ void k_quad_dopri<double, Topo>(const BatchArgs<double>, const AdaptiveArgs<double>, int);
#elif JM_CON_PART == 6 && JM_TOPO_QUAD
template __global__ void k_quad_dopri_gen<double, Topo>(const BatchArgs<double>, const AdaptiveArgs<double>, int);
#elif JM_CON_PART == 7 && JM_TOPO_QCON_SPLIT
template __global__ void k_quad_con_split<double, Topo, 1, 0>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 8 && JM_TOPO_QCON_SPLIT
template __global__ void k_quad_con_split<double, Topo, 2, 0>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 9 && JM_TOPO_QCON_SPLIT
template __global__ void k_qcon_pgs<double, Topo, 8, 0, JM_QCON_PGS_DEPTH>(const QConArgs<double>, const double *, unsigned);
template __global__ void k_qcon_pgs<double, Topo, 12, 64, JM_QCON_PGS_DEPTH - 1>(const QConArgs<double>, const double *, unsigned);
template __global__ void k_qcon_pgs_lane<double, Topo>(const QConArgs<double>, const double *, int32_t *);
#elif JM_CON_PART == 11 && JM_TOPO_QUAD
template __global__ void k_quad_con<double, Topo, 1>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 12 && JM_TOPO_QUAD
template __global__ void k_quad_con_gen<double, Topo, 1>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 13 && JM_TOPO_QCON_SPLIT
template __global__ void k_quad_con_split<double, Topo, 1, 1>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 14 && JM_TOPO_QCON_SPLIT
template __global__ void k_quad_con_split<double, Topo, 2, 1>(const BatchArgs<double>, const QConArgs<double>);
#elif JM_CON_PART == 15 && !JM_TOPO_QUAD
// the one-robot-per-lane kernels in the form that reads the applied wrenches (impulse / profile forces on frames, ABI 9):
// topologies without the branch-parallel structure (those have their variation kernels, parts 3 and 4)
template __global__ void k_batch<double, Topo, true>(const BatchArgs<double>);
#elif JM_CON_PART == 16 && !JM_TOPO_QUAD
template __global__ void k_constrained<double, Topo, true>(const BatchArgs<double>, const ConArgs<double>);
#elif JM_CON_PART == 10 && JM_TOPO_QCON_SPLIT
template __global__ void k_qtip_pgs<double, Topo>(const QConArgs<double>, const double *, unsigned);
template __global__ void k_qcon_exact<double, Topo>(const QConArgs<double>);
template __global__ void k_qtip_exact<double, Topo>(const QConArgs<double>);
#endif
}
