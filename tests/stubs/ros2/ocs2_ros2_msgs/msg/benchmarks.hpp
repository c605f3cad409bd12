// STAND-IN for the generated message headers of ocs2_ros2_msgs (Benchmarks.msg: float64 time, IndividualBenchmarks[] benchmarks;
// IndividualBenchmarks.msg: string description, float64[] values)
#pragma once
#include <string>
#include <vector>
namespace ocs2_ros2_msgs { namespace msg {
struct IndividualBenchmarks { std::string description; std::vector<double> values; };
struct Benchmarks { double time = 0.0; std::vector<IndividualBenchmarks> benchmarks; };
}}  // namespace ocs2_ros2_msgs::msg
