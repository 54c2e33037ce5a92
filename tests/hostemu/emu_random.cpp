// Host build of the sensor-delay sample selection of jiminy_amd/csrc/jm_random.h (tests only).
#define JM_HOST_EMU 1
#include <cstring>
#include "../../jiminy_amd/csrc/jm_random.h"

extern "C" void emu_delay_lookup(int n_hist, const double * times, int order, double cfg_delay, double cfg_jitter,
                                 double delay, int * idx, double * ratio)
{
    jm::DelayParams p;
    std::memset(&p, 0, sizeof(p));
    p.n_sensors = 1; p.n_fields = 1; p.n_hist = n_hist; p.order = order; p.has_history = 1;
    for (int i = 0; i < n_hist; ++i) { p.slot[i] = i; p.times[i] = times[i]; }
    p.delay[0] = cfg_delay; p.jitter[0] = (float)cfg_jitter;
    jm::delay_lookup(p, 0, delay, *idx, *ratio);
}

// the per-joint model-bias law of jm_block_model_bias (jm_random.h bias_one_joint) for one lane, on the host
extern "C" void emu_model_bias_lane(int njoints, int first, const double * nominal, const float * std4, uint64_t * state,
                                    double * out /* [13 * njoints] */)
{
    jm::rnd::ZigguratTables z;
    jm::rnd::ziggurat_tables(z);
    jm::BiasParams p{};
    p.njoints = njoints; p.first = first;
    p.inertia_std = std4[0]; p.mass_std = std4[1]; p.com_std = std4[2]; p.pos_std = std4[3];
    for (int j = first; j < njoints; ++j)
        jm::rnd::bias_one_joint(p, nominal + 25 * j, *state, z.kn, z.fn, z.wn, out + 13 * j);
}
