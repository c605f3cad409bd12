// STAND-IN for <ocs2_sqp/SqpSettings.h>: the fields of sqp::Settings the adaptor reads (names and defaults as upstream;
// g1_wb_mpc/config/mpc/task.info:79-93 overrides nThreads, dt, sqpIteration, deltaTol, g_max, g_min, ...)
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 { namespace sqp {
struct Settings {
  size_t sqpIteration = 1;
  scalar_t deltaTol = 1e-6, costTol = 1e-4;
  scalar_t alpha_decay = 0.5, alpha_min = 1e-4, gamma_c = 1e-6, g_max = 1e6, g_min = 1e-6, armijoFactor = 1e-4;
  scalar_t dt = 0.01;
  bool useFeedbackPolicy = true, projectStateInputEqualityConstraints = true, printSolverStatus = false, printSolverStatistics = false, printLinesearch = false;
  size_t nThreads = 4;
};
}}  // namespace ocs2::sqp
