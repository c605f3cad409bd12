// STAND-IN (test infrastructure) for <ocs2_mpc/SystemObservation.h>: upstream's POD {mode, time, state, input}; the compiled reference
// files include the header but take (time, state) arguments.
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 {
struct SystemObservation { size_t mode = 0; scalar_t time = 0.0; vector_t state; vector_t input; };
}  // namespace ocs2
