// STAND-IN (test infrastructure) for <ocs2_core/automatic_differentiation/Types.h>: upstream ad_scalar_t is
// CppAD::AD<CppAD::cg::CG<double>>; the files compiled here only name it in alias declarations.
#pragma once
#include <ocs2_core/Types.h>
namespace ocs2 { struct ad_scalar_t; }
