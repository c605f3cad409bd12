// C++ host side of the batch axis over the GPUs of one node: RAII over hsqp_comm_* (include/hsqp.h) — one process per GPU, a contiguous block of
// ceil(B / world) instances per rank, device buffers in and out.  The reference runs one instance on the host and has nothing of
// the kind; the shapes here are the ones its solver boundary moves (state / input trajectories and node parameters per instance:
// /root/reference/humanoid_nmpc/humanoid_wb_mpc/src/WBMpcInterface.cpp:113-121 hands them to ocs2::SqpMpc one instance at a time).
//
//   std::array<char, HSQP_COMM_ID_BYTES> id;                       // rank 0: HipSqpComm::uniqueId(id.data()), then shipped to the other processes
//   hsqp_host::HipSqpComm comm(id.data(), rank, world, device);
//   auto [lo, hi] = comm.shard(B);                                  // this rank's instances
//   comm.scatterRows(dGlobalX, dX, (N + 1) * HSQP_NX, B);           // ... one call per array of hsqp_problem, then hsqp_upload_device
//   solver.runDevice(...); comm.gatherRows(dX, dGlobalX, (N + 1) * HSQP_NX, B);
#pragma once
#include <stdexcept>
#include <string>
#include <utility>

#include "../../include/hsqp.h"

namespace hsqp_host {

class HipSqpComm {
 public:
  static void uniqueId(void* id) {
    if (const int rc = hsqp_comm_unique_id(id); rc != HSQP_OK) throw std::runtime_error(std::string("[HipSqpComm] ") + hsqp_comm_create_error() + " (" + std::to_string(rc) + ")");
  }
  HipSqpComm(const void* id, int rank, int world, int device) {
    if (const int rc = hsqp_comm_create(&c_, id, rank, world, device); rc != HSQP_OK)
      throw std::runtime_error(std::string("[HipSqpComm] ") + hsqp_comm_create_error() + " (" + std::to_string(rc) + ")");
  }
  ~HipSqpComm() { hsqp_comm_destroy(c_); }
  HipSqpComm(const HipSqpComm&) = delete;
  HipSqpComm& operator=(const HipSqpComm&) = delete;

  int rank() const { return hsqp_comm_rank(c_); }
  int world() const { return hsqp_comm_world(c_); }
  std::pair<int, int> shard(int globalBatch) const { int lo = 0, hi = 0; check(hsqp_comm_shard(c_, globalBatch, &lo, &hi)); return {lo, hi}; }
  static std::pair<int, int> shardOf(int globalBatch, int world, int rank) {
    int lo = 0, hi = 0;
    if (hsqp_comm_shard_of(globalBatch, world, rank, &lo, &hi) != HSQP_OK) throw std::runtime_error("[HipSqpComm] shardOf: bad argument");
    return {lo, hi};
  }
  void broadcast(void* dBuf, long long bytes, int root = 0) { check(hsqp_comm_broadcast(c_, dBuf, bytes, root)); }
  void scatterRows(const double* dGlobal, double* dLocal, long long rowDoubles, int globalBatch, int root = 0) {
    check(hsqp_comm_scatter_rows(c_, dGlobal, dLocal, rowDoubles, globalBatch, root));
  }
  void gatherRows(const double* dLocal, double* dGlobal, long long rowDoubles, int globalBatch, int root = 0) {
    check(hsqp_comm_gather_rows(c_, dLocal, dGlobal, rowDoubles, globalBatch, root));
  }
  void max(double* values, int n) { check(hsqp_comm_max(c_, values, n)); }
  void barrier() { check(hsqp_comm_barrier(c_)); }
  hsqp_comm* handle() { return c_; }

 private:
  void check(int rc) const {
    if (rc != HSQP_OK) throw std::runtime_error(std::string("[HipSqpComm] ") + hsqp_comm_last_error(c_) + " (" + std::to_string(rc) + ")");
  }
  hsqp_comm* c_ = nullptr;
};

}  // namespace hsqp_host
