// =====================================================================================
// TEST INFRASTRUCTURE — C entry points over the REFERENCE'S OWN swing-foot planner and gait schedule, compiled from
// /root/reference in place (oracle/Makefile, target _ref/libref_swing.so; stand-ins for the absent third-party headers in
// oracle/ref_stubs/).  Only tests/ and tests/golden/make_ref_swing_golden.py load the library.  It pins SURVEY.md §8 rows
// a16/a17 (host mirror wb_humanoid_mpc_amd/reference.py and device kernel k_params) against reference-compiled code:
//   humanoid_common_mpc/src/swing_foot_planner/CubicSpline.cpp:38-80, SplineCpg.cpp:38-62,
//   SwingTrajectoryPlanner.cpp:52-191 (update, getZ*Constraint, getImpactProximityFactor),
//   humanoid_common_mpc/src/gait/GaitSchedule.cpp:48-145 (insertModeSequenceTemplate, getModeSchedule, tileModeSequenceTemplate),
//   humanoid_common_mpc/include/humanoid_common_mpc/gait/MotionPhaseDefinition.h:58-76 (modeNumber2StanceLeg).
// =====================================================================================
#include <cstring>

#include "humanoid_common_mpc/gait/GaitSchedule.h"
#include "humanoid_common_mpc/gait/MotionPhaseDefinition.h"
#include "humanoid_common_mpc/swing_foot_planner/SwingTrajectoryPlanner.h"

using namespace ocs2;
using namespace ocs2::humanoid;

// declared by the reference's headers, defined in files that need Boost for real: never called by this driver
namespace ocs2::humanoid {
ModeSchedule loadModeSchedule(const std::string&, const std::string&, bool) { throw std::runtime_error("stand-in"); }
ModeSequenceTemplate loadModeSequenceTemplate(const std::string&, const std::string&, bool) { throw std::runtime_error("stand-in"); }
std::ostream& operator<<(std::ostream& stream, const ModeSequenceTemplate&) { return stream; }
}  // namespace ocs2::humanoid

extern "C" {

// CubicSpline(start, end): out[n][3] = position, velocity, acceleration at t[i]
void ref_cubic_spline(const double start[3], const double end[3], int n, const double* t, double* out) {
  const CubicSpline s(CubicSpline::Node{start[0], start[1], start[2]}, CubicSpline::Node{end[0], end[1], end[2]});
  for (int i = 0; i < n; ++i) { out[3 * i] = s.position(t[i]); out[3 * i + 1] = s.velocity(t[i]); out[3 * i + 2] = s.acceleration(t[i]); }
}

void ref_spline_cpg(const double lift[3], double mid, const double touch[3], int n, const double* t, double* out) {
  const SplineCpg s(CubicSpline::Node{lift[0], lift[1], lift[2]}, mid, CubicSpline::Node{touch[0], touch[1], touch[2]});
  for (int i = 0; i < n; ++i) { out[3 * i] = s.position(t[i]); out[3 * i + 1] = s.velocity(t[i]); out[3 * i + 2] = s.acceleration(t[i]); }
}

void ref_mode_to_stance_legs(int mode, int flags[2]) {
  const contact_flag_t f = modeNumber2StanceLeg(static_cast<size_t>(mode));
  flags[0] = f[0]; flags[1] = f[1];
}

// cfg = {liftOffVelocity, touchDownVelocity, swingHeight, touchDownHeightOffset, swingTimeScale, impactProximityFactorMidPointValue,
//        impactProximityFactorLiftOffVelocity, impactProximityFactorTouchDownVelocity} (the order of hsqp_swing_config).
// out[n][2][4] = per leg {z, zdot, zddot, impact proximity}; mode_out[n] = modeAtTime.  Returns 0, or 1 if update() throws
// (a swing phase without lift-off / touch-down inside the schedule).
int ref_swing_planner(const double cfg[8], int n_events, const double* event_times, const int* mode_sequence, double terrain_height, int n,
                      const double* t, double* out, int* mode_out) {
  SwingTrajectoryPlanner::Config c;
  c.liftOffVelocity = cfg[0]; c.touchDownVelocity = cfg[1]; c.swingHeight = cfg[2]; c.touchDownHeightOffset = cfg[3]; c.swingTimeScale = cfg[4];
  c.impactProximityFactorMidPointValue = cfg[5]; c.impactProximityFactorLiftOffVelocity = cfg[6]; c.impactProximityFactorTouchDownVelocity = cfg[7];
  ModeSchedule ms(std::vector<scalar_t>(event_times, event_times + n_events), std::vector<size_t>(mode_sequence, mode_sequence + n_events + 1));
  SwingTrajectoryPlanner planner(c, 2);
  try {
    planner.update(ms, terrain_height);
  } catch (const std::exception&) {
    return 1;
  }
  for (int i = 0; i < n; ++i)
    for (int leg = 0; leg < 2; ++leg) {
      double* o = out + (2 * i + leg) * 4;
      o[0] = planner.getZpositionConstraint(leg, t[i]);
      o[1] = planner.getZvelocityConstraint(leg, t[i]);
      o[2] = planner.getZaccelerationConstraint(leg, t[i]);
      o[3] = planner.getImpactProximityFactor(leg, t[i]);
    }
  if (mode_out) for (int i = 0; i < n; ++i) mode_out[i] = static_cast<int>(ms.modeAtTime(t[i]));
  return 0;
}

// GaitSchedule(initialModeSchedule {[0.5], [STANCE, STANCE]} as reference.info, defaultTemplate STANCE [0, 0.5], phaseTransitionStanceTime)
// -> insertModeSequenceTemplate(template, start, final) -> getModeSchedule(lower, upper).
// Returns the number of events written (<= cap), or -1 if the reference throws / cap is too small.
int ref_gait_schedule(int n_tpl, const double* tpl_times /*[n_tpl+1]*/, const int* tpl_modes /*[n_tpl]*/, double phase_transition_stance_time,
                      double insert_start, double insert_final, double lower, double upper, int cap, double* event_times, int* mode_sequence) {
  try {
    GaitSchedule gs(ModeSchedule({0.5}, {STANCE, STANCE}), ModeSequenceTemplate({0.0, 0.5}, {STANCE}), phase_transition_stance_time);
    gs.insertModeSequenceTemplate(ModeSequenceTemplate(std::vector<scalar_t>(tpl_times, tpl_times + n_tpl + 1), std::vector<size_t>(tpl_modes, tpl_modes + n_tpl)),
                                  insert_start, insert_final);
    const ModeSchedule ms = gs.getModeSchedule(lower, upper);
    const int ne = static_cast<int>(ms.eventTimes.size());
    if (ne > cap) return -1;
    for (int i = 0; i < ne; ++i) event_times[i] = ms.eventTimes[i];
    for (int i = 0; i <= ne; ++i) mode_sequence[i] = static_cast<int>(ms.modeSequence[i]);
    return ne;
  } catch (const std::exception&) {
    return -1;
  }
}

}  // extern "C"
