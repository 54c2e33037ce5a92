// jm_pack.h -- host side: validate a jm_model_desc against the compiled-in topology and pack
// the numeric parameters into the block layout the kernels read (Layout<Tp> in jm_kernels.h).
#pragma once
#include <cmath>
#include <string>
#include <vector>

#include "jm_kernels.h"
#include "jm_quad.h"

namespace jm
{
template<class Tp> inline bool check_topology(const jm_model_desc & d, std::string & why)
{
    auto fail = [&](const char * what) { why = std::string("model topology mismatch: ") + what; return false; };
    if (d.njoints != Tp::NJ || d.nq != Tp::NQ || d.nv != Tp::NV) return fail("joint/dof counts");
    if (d.nmotors != Tp::NM || d.ncontacts != Tp::NC) return fail("motor/contact counts");
    if (d.nimu != Tp::NIMU || d.nforce != Tp::NFORCE || d.ncontact_sensors != Tp::NCS ||
        d.nencoder != Tp::NENC || d.neffort != Tp::NEFF) return fail("sensor counts");
    for (int j = 0; j < Tp::NJ; ++j)
    {
        if (d.parents[j] != Tp::parent[j]) return fail("parents");
        if (d.jtypes[j] != Tp::jtype[j]) return fail("joint types");
        if (d.idx_q[j] != Tp::idx_q[j] || d.idx_v[j] != Tp::idx_v[j]) return fail("index maps");
    }
    for (int m = 0; m < Tp::NM; ++m)
        if (d.motor_joint[m] != Tp::motor_joint[m] || d.motor_flags[m] != Tp::motor_flags[m]) return fail("motors");
    for (int c = 0; c < Tp::NC; ++c)
        if (d.contact_joint[c] != Tp::contact_joint[c]) return fail("contact frames");
    for (int s = 0; s < Tp::NIMU; ++s)
        if (d.imu_joint[s] != Tp::imu_joint[s]) return fail("imu frames");
    for (int s = 0; s < Tp::NFORCE; ++s)
        if (d.force_joint[s] != Tp::force_joint[s]) return fail("force sensor frames");
    for (int s = 0; s < Tp::NCS; ++s)
        if (d.contact_sensor_contact[s] != Tp::cs_contact[s]) return fail("contact sensors");
    for (int s = 0; s < Tp::NENC; ++s)
        if (d.encoder_joint[s] != Tp::enc_joint[s] || (d.encoder_joint_side[s] != 0) != (Tp::enc_side[s] != 0))
            return fail("encoders");
    for (int s = 0; s < Tp::NEFF; ++s)
        if (d.effort_motor[s] != Tp::eff_motor[s]) return fail("effort sensors");
    if (d.n_constraint_frames != Tp::NX) return fail("user constraint frame count");
    for (int x = 0; x < Tp::NX; ++x)
        if (d.cframe_joint[x] != Tp::xframe_joint[x] || d.cframe_mask[x] != Tp::xframe_mask[x] || d.cframe_kind[x] != Tp::xframe_kind[x] ||
            d.cframe_joint2[x] != Tp::xframe_joint2[x]) return fail("user constraint frames");
    if (d.n_constraint_joints != Tp::NXJ) return fail("user constraint joint count");
    for (int k = 0; k < Tp::NXJ; ++k)
        if (d.cjoint_joint[k] != Tp::xjoint[k]) return fail("user constraint joints");
    return true;
}

// model part of the parameter block (everything except the options tail)
template<class Tp> inline std::vector<double> pack_model(const jm_model_desc & d)
{
    using L = Layout<Tp>;
    std::vector<double> P(L::TOTAL, 0.0);
    for (int j = 0; j < Tp::NJ; ++j)
    {
        double * o = &P[L::JOINT + j * L::JSTRIDE];
        for (int k = 0; k < 9; ++k) o[k] = d.placement_R[9 * j + k];
        for (int k = 0; k < 3; ++k) o[9 + k] = d.placement_p[3 * j + k];
        o[12] = d.mass[j];
        for (int k = 0; k < 3; ++k) o[13 + k] = d.com[3 * j + k];
        const double * I = d.inertia + 9 * j;
        o[16] = I[0]; o[17] = I[1]; o[18] = I[2]; o[19] = I[4]; o[20] = I[5]; o[21] = I[8];
        for (int k = 0; k < 3; ++k) o[22 + k] = d.axes[3 * j + k];
    }
    for (int i = 0; i < Tp::NV; ++i) P[L::ROTOR + i] = d.rotor_inertia[i];
    for (int i = 0; i < Tp::NQ; ++i) { P[L::QLO + i] = d.position_lower[i]; P[L::QHI + i] = d.position_upper[i]; }
    for (int m = 0; m < Tp::NM; ++m)
        for (int k = 0; k < JM_MOTOR_NPARAMS; ++k) P[L::MOTOR + JM_MOTOR_NPARAMS * m + k] = d.motor_params[JM_MOTOR_NPARAMS * m + k];
    auto put_frame = [&](int off, const double * R, const double * p) {
        for (int k = 0; k < 9; ++k) P[off + k] = R[k];
        for (int k = 0; k < 3; ++k) P[off + 9 + k] = p[k];
    };
    for (int c = 0; c < Tp::NC; ++c) put_frame(L::CONTACT + 12 * c, d.contact_R + 9 * c, d.contact_p + 3 * c);
    for (int s = 0; s < Tp::NIMU; ++s) put_frame(L::IMU + 12 * s, d.imu_R + 9 * s, d.imu_p + 3 * s);
    // force sensor <- contact relative placements: F^-1 * C (basic_sensors.cc:326-350)
    for (int s = 0; s < Tp::NFORCE; ++s)
        for (int c = 0; c < Tp::NC; ++c)
        {
            if (d.contact_joint[c] != d.force_joint[s]) continue;
            const double * RF = d.force_R + 9 * s, * pF = d.force_p + 3 * s;
            const double * RC = d.contact_R + 9 * c, * pC = d.contact_p + 3 * c;
            double R[9], p[3];
            for (int i = 0; i < 3; ++i)
            {
                for (int k = 0; k < 3; ++k)
                    R[3 * i + k] = RF[0 + i] * RC[0 + k] + RF[3 + i] * RC[3 + k] + RF[6 + i] * RC[6 + k];
                p[i] = RF[0 + i] * (pC[0] - pF[0]) + RF[3 + i] * (pC[1] - pF[1]) + RF[6 + i] * (pC[2] - pF[2]);
            }
            put_frame(L::FREL + 12 * (s * Tp::NC + c), R, p);
        }
    for (int s = 0; s < Tp::NENC; ++s) P[L::ENC + s] = d.encoder_reduction[s];
    for (int x = 0; x < Tp::NX; ++x)
    {
        put_frame(L::XFRAME + 12 * x, d.cframe_R + 9 * x, d.cframe_p + 3 * x);
        for (int k = 0; k < 8; ++k) P[L::XPAR + 8 * x + k] = d.cframe_params[8 * x + k];
    }
    // flexibility of the spherical joints: stiffness 3, damping 3 (zero without a configuration)
    for (int j = 0, k = 0; j < Tp::NJ; ++j)
        if (Tp::jtype[j] == JM_JT_SPHERICAL)
        {
            for (int i = 0; i < 3; ++i)
            {
                P[L::FLEX + 6 * k + i] = d.flex_stiffness ? d.flex_stiffness[3 * j + i] : 0.0;
                P[L::FLEX + 6 * k + 3 + i] = d.flex_damping ? d.flex_damping[3 * j + i] : 0.0;
            }
            ++k;
        }
    return P;
}
// total size of the parameter block, including the limb table of the limb-parallel kernel
template<class Tp> constexpr int param_total()
{
    if constexpr (Tp::QUAD) return QLayout<Tp>::TOTAL;
    else return Layout<Tp>::TOTAL;
}
// limb table of the branch-parallel kernel (jm_quad.h QLayout), appended to the parameter block
template<class Tp> inline void pack_quad(std::vector<double> & P, const jm_model_desc & d)
{
    if constexpr (Tp::QUAD)
    {
        using Q = QLayout<Tp>;
        P.resize(Q::TOTAL, 0.0);
        for (int k = 0; k < 4; ++k)
        {
            double * T = &P[Q::OFFSET + k * Q::QSTRIDE];
            for (int s = 0; s < Tp::QN; ++s)
            {
                const int j = Tp::limb_joint[k][s];
                double * o = T + s * Q::QJ;
                if (j < 0)
                {
                    // dummy joint padding a short limb at its tip: identity placement, no mass, unit
                    // rotor inertia (D = 1), no motor, unbounded: it never moves and transmits the
                    // wrench of the contact points (attached to the padded tip) unchanged
                    o[Q::J_PLC + 0] = o[Q::J_PLC + 4] = o[Q::J_PLC + 8] = 1.0;
                    o[Q::J_AXIS] = 1.0;
                    o[Q::J_AXP] = 1.0;
                    o[Q::J_ROTOR] = 1.0;
                    o[Q::J_QLO] = -1.0e300;
                    o[Q::J_QHI] = 1.0e300;
                    o[Q::J_ENC] = 1.0;
                    continue;
                }
                for (int i = 0; i < 9; ++i) o[Q::J_PLC + i] = d.placement_R[9 * j + i];
                for (int i = 0; i < 3; ++i) o[Q::J_PLC + 9 + i] = d.placement_p[3 * j + i];
                o[Q::J_RBI] = d.mass[j];
                for (int i = 0; i < 3; ++i) o[Q::J_RBI + 1 + i] = d.com[3 * j + i];
                const double * I = d.inertia + 9 * j;
                o[Q::J_RBI + 4] = I[0]; o[Q::J_RBI + 5] = I[1]; o[Q::J_RBI + 6] = I[2];
                o[Q::J_RBI + 7] = I[4]; o[Q::J_RBI + 8] = I[5]; o[Q::J_RBI + 9] = I[8];
                // explicit unit axis also for axis-aligned joints (the limb code is type-generic)
                const int t = d.jtypes[j];
                double ax[3] = {d.axes[3 * j], d.axes[3 * j + 1], d.axes[3 * j + 2]};
                if (t == JM_JT_RX) { ax[0] = 1; ax[1] = 0; ax[2] = 0; }
                if (t == JM_JT_RY) { ax[0] = 0; ax[1] = 1; ax[2] = 0; }
                if (t == JM_JT_RZ) { ax[0] = 0; ax[1] = 0; ax[2] = 1; }
                for (int i = 0; i < 3; ++i)
                {
                    o[Q::J_AXIS + i] = ax[i];
                    // axis seen from the parent joint frame: placement rotation times axis
                    o[Q::J_AXP + i] = d.placement_R[9 * j + 3 * i] * ax[0] + d.placement_R[9 * j + 3 * i + 1] * ax[1]
                                      + d.placement_R[9 * j + 3 * i + 2] * ax[2];
                }
                o[Q::J_ROTOR] = d.rotor_inertia[d.idx_v[j]];
                o[Q::J_QLO] = d.position_lower[d.idx_q[j]];
                o[Q::J_QHI] = d.position_upper[d.idx_q[j]];
                const int m = Tp::limb_motor[k][s];
                for (int i = 0; i < JM_MOTOR_NPARAMS; ++i) o[Q::J_MOTOR + i] = d.motor_params[JM_MOTOR_NPARAMS * m + i];
                o[Q::J_ENC] = Tp::QHAS_ENC ? d.encoder_reduction[Tp::limb_enc[k][s]] : 1.0;
            }
            for (int c = 0; c < Tp::QCL; ++c)
            {
                const int ci = Tp::limb_contact[k][c];
                double * o = T + Q::CONTACT + c * Q::QC;
                if (ci < 0)
                {
                    o[0] = o[4] = o[8] = 1.0;
                    continue;
                }
                for (int i = 0; i < 9; ++i) o[i] = d.contact_R[9 * ci + i];
                for (int i = 0; i < 3; ++i) o[9 + i] = d.contact_p[3 * ci + i];
                o[Q::C_IDX] = (double)ci;
                o[Q::C_CS] = (double)(Tp::limb_cs[k][c] < 0 ? 0 : Tp::limb_cs[k][c]);
                if constexpr (Tp::QHAS_FORCE)
                {
                    const int fs = Tp::limb_force[k];
                    if (fs >= 0)
                    {
                        const double * src = &P[Layout<Tp>::FREL + 12 * (fs * Tp::NC + ci)];
                        for (int i = 0; i < 12; ++i) o[Q::C_FREL + i] = src[i];
                    }
                }
            }
        }
    }
    else { (void)P; (void)d; }
}
template<class Tp> inline void pack_options(std::vector<double> & P, const jm_options & o)
{
    using L = Layout<Tp>;
    for (int k = 0; k < 6; ++k) P[L::OPT + k] = o.gravity[k];
    P[L::OPT + 6] = o.contact_stiffness;
    P[L::OPT + 7] = o.contact_damping;
    P[L::OPT + 8] = o.contact_friction;
    P[L::OPT + 9] = o.contact_transition_eps;
    P[L::OPT + 10] = o.contact_transition_velocity;
}
inline jm_options default_options()
{
    jm_options o;
    o.gravity[0] = 0; o.gravity[1] = 0; o.gravity[2] = -9.81; o.gravity[3] = 0; o.gravity[4] = 0; o.gravity[5] = 0;
    o.contact_stiffness = 1.0e6;
    o.contact_damping = 2.0e3;
    o.contact_friction = 1.0;
    o.contact_transition_eps = 1.0e-3;
    o.contact_transition_velocity = 1.0e-2;
    return o;
}
}  // namespace jm
