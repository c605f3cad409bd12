// STAND-IN (test infrastructure) for <cppad/cg.hpp>: nothing of CppAD is used by the files compiled here (ad_scalar_t is the number wrapper of ref_stubs/ocs2_core/automatic_differentiation/Types.h)
#pragma once
