// STAND-IN for <rclcpp/rclcpp.hpp>: the names SqpBenchmarksPublisher.cpp touches.  The publisher keeps the last message so that a test can read it.
#pragma once
#include <memory>
#include <string>
namespace rclcpp {
class QoS { public: explicit QoS(size_t) {} QoS& best_effort() { return *this; } };
template <class Msg> class Publisher {
 public:
  using SharedPtr = std::shared_ptr<Publisher<Msg>>;
  void publish(const Msg& m) { last = m; ++count; }
  Msg last; int count = 0;
};
class Node {
 public:
  using SharedPtr = std::shared_ptr<Node>;
  template <class Msg> typename Publisher<Msg>::SharedPtr create_publisher(const std::string& topic_, const QoS&) { topic = topic_; auto p = std::make_shared<Publisher<Msg>>(); last_publisher = p; return p; }
  std::string topic;
  std::shared_ptr<void> last_publisher;
};
}  // namespace rclcpp
