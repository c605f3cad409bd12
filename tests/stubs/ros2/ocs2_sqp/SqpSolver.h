// The integration recipe: code written against ocs2::SqpSolver — the reference's SqpBenchmarksPublisher takes a `const ocs2::SqpSolver*`
// and reads `SqpSolver::Benchmarks` — builds against the HIP adaptor through this one alias.
#pragma once
#include "HipSqpSolverAdaptor.h"
namespace ocs2 { using SqpSolver = humanoid::HipSqpSolverAdaptor; }
