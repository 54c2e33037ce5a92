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
