// MOCK (test infrastructure): see pinocchio/mock.hpp
#pragma once
#include <pinocchio/mock.hpp>
