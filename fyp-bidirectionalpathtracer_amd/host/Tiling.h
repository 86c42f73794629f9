// Tiling.h — one rank's end of the multi-GPU tiling of the frame over RCCL.
//
// The reference renders on one GPU: RayLaunch::execute ends in a single DispatchRays over the whole frame
// (SharedUtils/RayLaunch.cpp:200-223 -> Falcor API/D3D12/D3D12RenderContext.cpp:350-384).  Here the frame's rows are
// dealt to `world` ranks in interleaved stripes (bdpt_resize_stripes), every rank holds the scene and its BVH, and
// the ONE exchange step of the path is the light-tracing splat buffer:
//
//   bdpt_execute(DEFER_TAIL | DEFER_RESOLVE)      everything that writes the splat accumulators
//   TileExchange::reduceScatter                   ncclReduceScatter(ncclUint64, ncclSum) of the owner-major buffer on
//                                                 the exchange stream, ordered behind the render stream by an event
//   bdpt_execute_tail                             the zero-valued connection rounds, beside the exchange
//   TileExchange::waitFor + bdpt_resolve_tile     the rank's own chunk, summed over all ranks, folded into its rows
//   bdpt_accumulate_tile                          running mean over its rows
//   TileExchange::allGather                       tile framebuffers -> the frame, only when an image is read back
//
// Integer sums are exact and order-independent, so N ranks produce the bits one rank produces.
// RCCL is linked by the HOST (this file); libbdpt_amd.so and its C ABI know nothing about it.
#pragma once
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace bdpt {

class TileExchange {
 public:
  using SharedPtr = std::shared_ptr<TileExchange>;
  static constexpr uint32_t kMaxSlots = 8;  // frames in flight a pipeline may keep (RenderingPipeline::setFramesInFlight)

  // `comm` is this rank's communicator (ncclCommInitAll / ncclCommInitRank by the host program, which also destroys it);
  // nullptr with world == 1 means "no exchange": the rank's chunk is the buffer itself.
  static SharedPtr create(int device, uint32_t rank, uint32_t world, ncclComm_t comm);
  ~TileExchange();

  uint32_t rank() const { return mRank; }
  uint32_t world() const { return mWorld; }
  // Sum `full` (world * chunkU64 words, owner-major) over the ranks; this rank's chunk lands in `mine`.  Enqueued on the
  // exchange stream behind everything `renderStream` holds so far; returns at once.  `slot` names the frame slot (one
  // event pair each, so frames in flight do not wait on one another's exchange).
  bool reduceScatter(const uint64_t* full, uint64_t* mine, uint64_t chunkU64, hipStream_t renderStream, uint32_t slot);
  // renderStream continues only after slot's exchange has landed
  void waitFor(hipStream_t renderStream, uint32_t slot);
  // every rank's `count` floats -> all (world * count floats, rank-major), on `stream`
  bool allGather(const float* mine, float* all, size_t count, hipStream_t stream);
  const std::string& lastError() const { return mError; }
  // This rank cannot take part in a collective its peers are about to enter (a launch or an allocation failed on it):
  // ncclCommAbort on its communicator, so that the peers' collectives return with an error instead of waiting for
  // ever.  Every later reduceScatter / allGather of this object fails at once.  (No communicator: nothing to do.)
  // A host program that keeps several communicators (ranks as threads) or has its own teardown installs a handler: it is
  // called INSTEAD of ncclCommAbort and must see to it that every communicator of the job is aborted exactly once
  // (bdpt_render: RankGroup::abort -> the watcher).
  void abort(const std::string& why);
  bool aborted() const { return mAborted; }
  void setAbortHandler(std::function<void()> f) { mOnAbort = std::move(f); }

 private:
  TileExchange() = default;
  int mDevice = 0;
  uint32_t mRank = 0, mWorld = 1;
  ncclComm_t mComm = nullptr;
  hipStream_t mStream = nullptr;  // the exchange stream
  hipEvent_t mReady[kMaxSlots] = {}, mDone[kMaxSlots] = {};
  std::string mError;
  bool mAborted = false;
  std::function<void()> mOnAbort;
};

// Rows per stripe for a frame of `height` rows dealt to `world` ranks: bdpt_stripe_rows (include/bdpt.h), the same rule
// the Python host uses (tiling.stripe_rows).
uint32_t stripeRows(uint32_t height, uint32_t world);

// Multi-process hosts (one process per GPU, e.g. under mpirun or a job scheduler): rank 0 writes the ncclUniqueId to
// `path`, the others wait for it; false after `timeoutSeconds`.  The file is single-use and carries `nonce` (RankSync.h
// "id file": rank 0 removes what it finds at `path` first, the others only take a file with their own nonce, and rank 0
// calls retireUniqueIdFile once ncclCommInitRank has returned — by then every rank has read it).  A launcher that
// re-runs a job with the same path gives each run its own nonce (bdpt_render --job-id) and is then safe against the
// file a crashed earlier run left behind; with nonce 0 only the single-use rule protects.
bool exchangeUniqueIdThroughFile(const std::string& path, uint32_t rank, ncclUniqueId* id, double timeoutSeconds = 120.0, uint64_t nonce = 0);
void retireUniqueIdFile(const std::string& path);

}  // namespace bdpt
