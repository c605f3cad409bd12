// STAND-IN (test infrastructure) for <ocs2_core/thread_support/Synchronized.h>: only included, never used by the files compiled here.
#pragma once
namespace ocs2 { template <class T> class Synchronized {}; }
