// STAND-IN (test infrastructure): included by a reference file, nothing of it is used
#pragma once
