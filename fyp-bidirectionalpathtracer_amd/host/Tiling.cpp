// Tiling.cpp — RCCL side of the tiled frame loop (Tiling.h).
#include "Tiling.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>

#include "../../include/bdpt.h"
#include "RankSync.h"

namespace bdpt {

uint32_t stripeRows(uint32_t height, uint32_t world) { return bdpt_stripe_rows(height, world); }

TileExchange::SharedPtr TileExchange::create(int device, uint32_t rank, uint32_t world, ncclComm_t comm) {
  if (world == 0 || rank >= world || (world > 1 && !comm)) return nullptr;
  SharedPtr x(new TileExchange());
  x->mDevice = device;
  x->mRank = rank;
  x->mWorld = world;
  x->mComm = comm;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  if (hipStreamCreateWithFlags(&x->mStream, hipStreamNonBlocking) != hipSuccess) return nullptr;
  for (uint32_t i = 0; i < kMaxSlots; i++)
    if (hipEventCreateWithFlags(&x->mReady[i], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&x->mDone[i], hipEventDisableTiming) != hipSuccess)
      return nullptr;
  return x;
}

TileExchange::~TileExchange() {
  if (mStream) {
    (void)hipStreamSynchronize(mStream);
    (void)hipStreamDestroy(mStream);
  }
  for (uint32_t i = 0; i < kMaxSlots; i++) {
    if (mReady[i]) (void)hipEventDestroy(mReady[i]);
    if (mDone[i]) (void)hipEventDestroy(mDone[i]);
  }
}

void TileExchange::abort(const std::string& why) {
  if (mAborted) return;
  mAborted = true;
  mError = "aborted: " + why;
  std::fprintf(stderr, "[TileExchange] rank %u leaves the group: %s\n", mRank, why.c_str());
  if (mOnAbort)
    mOnAbort();  // the host program aborts the job's communicators (each once)
  else if (mComm)
    (void)ncclCommAbort(mComm);  // (the host program owns the communicator and must not destroy it again)
  mComm = nullptr;
}

bool TileExchange::reduceScatter(const uint64_t* full, uint64_t* mine, uint64_t chunkU64, hipStream_t renderStream, uint32_t slot) {
  if (mAborted) return false;
  if (slot >= kMaxSlots || !full || !mine) return false;
  if (hipEventRecord(mReady[slot], renderStream) != hipSuccess) return false;
  if (hipStreamWaitEvent(mStream, mReady[slot], 0) != hipSuccess) return false;
  if (mComm) {
    const ncclResult_t r = ncclReduceScatter(full, mine, (size_t)chunkU64, ncclUint64, ncclSum, mComm, mStream);
    if (r != ncclSuccess) {
      mError = std::string("ncclReduceScatter: ") + ncclGetErrorString(r);
      return false;
    }
  } else if (mine != full + (size_t)mRank * chunkU64) {  // one rank, no communicator: its chunk is the buffer
    if (hipMemcpyAsync(mine, full + (size_t)mRank * chunkU64, (size_t)chunkU64 * 8, hipMemcpyDeviceToDevice, mStream) != hipSuccess) return false;
  }
  return hipEventRecord(mDone[slot], mStream) == hipSuccess;
}

void TileExchange::waitFor(hipStream_t renderStream, uint32_t slot) {
  if (slot < kMaxSlots) (void)hipStreamWaitEvent(renderStream, mDone[slot], 0);
}

bool TileExchange::allGather(const float* mine, float* all, size_t count, hipStream_t stream) {
  if (mAborted) return false;
  if (mComm) {
    const ncclResult_t r = ncclAllGather(mine, all, count, ncclFloat32, mComm, stream);
    if (r != ncclSuccess) {
      mError = std::string("ncclAllGather: ") + ncclGetErrorString(r);
      return false;
    }
    return true;
  }
  return hipMemcpyAsync(all + (size_t)mRank * count, mine, count * 4, hipMemcpyDeviceToDevice, stream) == hipSuccess;
}

bool exchangeUniqueIdThroughFile(const std::string& path, uint32_t rank, ncclUniqueId* id, double timeoutSeconds, uint64_t nonce) {
  if (rank == 0) {
    if (ncclGetUniqueId(id) != ncclSuccess) return false;
    return writeIdFile(path, nonce, id, sizeof(*id));
  }
  return waitForIdFile(path, nonce, id, sizeof(*id), timeoutSeconds);
}
void retireUniqueIdFile(const std::string& path) { retireIdFile(path); }

}  // namespace bdpt
