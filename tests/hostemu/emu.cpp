// Host emulation of the per-lane kernel code (tests only, never part of the product path):
// compiles jiminy_amd/csrc/jm_kernels.h with g++ (-DJM_HOST_EMU) so that the kernel logic can be
// compared with the oracle on machines without a GPU.
#define JM_HOST_EMU 1
#include <cstring>
#include <string>
#include <vector>
#include JM_TOPO_HEADER
#include "../../jiminy_amd/csrc/jm_kernels.h"
#include "../../jiminy_amd/csrc/jm_pack.h"

extern "C"
{
struct emu_io
{
    long long B;
    void *q, *v, *a, *command, *u_motor, *u, *f_external, *contact_forces, *imu, *force, *contact,
        *encoder, *effort, *energy, *joint_forces, *centroidal, *status, *q_in, *v_in, *a_out, *mask, *q_init, *v_init;
};
const char * emu_signature() { return Topo::signature; }
}

template<class T>
static int run(const jm_model_desc * d, const jm_options * o, const emu_io * io, int mode, int solver, double dt,
               int n_sub, int command_changed, int update_sensors)
{
    std::string why;
    if (!jm::check_topology<Topo>(*d, why)) return JM_ETOPOLOGY;
    std::vector<double> Pd = jm::pack_model<Topo>(*d);
    jm::pack_options<Topo>(Pd, *o);
    std::vector<T> P(Pd.begin(), Pd.end());
    jm::BatchArgs<T> A;
    std::memset(&A, 0, sizeof(A));
    A.P = P.data();
    A.q = (T *)io->q; A.v = (T *)io->v; A.a = (T *)io->a; A.command = (const T *)io->command;
    A.u_motor = (T *)io->u_motor; A.u = (T *)io->u; A.f_external = (T *)io->f_external;
    A.contact_forces = (T *)io->contact_forces; A.imu = (T *)io->imu; A.force = (T *)io->force;
    A.contact = (T *)io->contact; A.encoder = (T *)io->encoder; A.effort = (T *)io->effort;
    A.energy = (T *)io->energy; A.joint_forces = (T *)io->joint_forces; A.centroidal = (T *)io->centroidal;
    A.status = (int32_t *)io->status;
    A.q_in = (const T *)io->q_in; A.v_in = (const T *)io->v_in; A.a_out = (T *)io->a_out;
    A.mask = (const unsigned char *)io->mask; A.q_init = (const T *)io->q_init; A.v_init = (const T *)io->v_init;
    A.B = io->B; A.mode = mode; A.solver = solver; A.n_sub = n_sub; A.command_changed = command_changed;
    A.update_sensors = update_sensors; A.dt = (T)dt;
    std::vector<T> sb(jm::stage_rows<Topo>() + 1);
    for (long long lane = 0; lane < io->B; ++lane) jm::lane_run<T, Topo, 1>(A, lane, sb.data());
    return 0;
}
extern "C" int emu_run(const jm_model_desc * d, const jm_options * o, const emu_io * io, int dtype, int mode, int solver,
            double dt, int n_sub, int command_changed, int update_sensors)
{
    if (dtype == JM_F64) return run<double>(d, o, io, mode, solver, dt, n_sub, command_changed, update_sensors);
    return run<float>(d, o, io, mode, solver, dt, n_sub, command_changed, update_sensors);
}
