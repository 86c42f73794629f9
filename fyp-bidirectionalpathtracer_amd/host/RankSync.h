// RankSync.h — host-side logic of the multi-GPU frame loop that needs no GPU to be checked (tests/host_compile/
// host_logic_test.cpp runs all of it on the CPU):
//
//   stripe maths     which rows a rank renders, how its rows are packed for the all-gather and where a packed row goes
//                    back to (the rule of bdpt_resize_stripes / tiling.stripes_of, in one place for readOutput and the
//                    tiled denoiser);
//   id file          how rank 0 hands the ncclUniqueId to the other PROCESSES of a job: a single-use file that carries
//                    the job's nonce, so a file left behind by an earlier run is never taken for this run's;
//   RankGroup        the ranks of ONE process (threads): a barrier that can be aborted, and an agreement point — every
//                    rank says whether its set-up succeeded and all of them leave together when one did not, instead
//                    of the healthy ranks waiting for ever in a barrier or a collective the failed rank never enters.
//
// Header-only, standard library only.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace bdpt {

// ---- stripes ------------------------------------------------------------------------------------------------------
// Rows are dealt to `world` ranks in stripes of R rows: rank r renders stripes r, r + world, ...  A rank's PACKED rows
// are its stripes in ascending order; every rank's packed buffer is padded to chunkRows (the same for all ranks).
struct StripeSpan {
  uint32_t firstRow;   // frame row the stripe starts at
  uint32_t rows;       // its height (the frame's last stripe may be short)
  uint32_t packedRow;  // row of the rank's packed buffer it starts at
};
inline uint32_t stripeChunkRows(uint32_t height, uint32_t world, uint32_t R) {
  const uint32_t n = (height + R - 1) / R;
  return ((n + world - 1) / world) * R;
}
inline std::vector<StripeSpan> stripeSpans(uint32_t height, uint32_t world, uint32_t rank, uint32_t R) {
  std::vector<StripeSpan> v;
  const uint32_t n = (height + R - 1) / R;
  uint32_t at = 0;
  for (uint32_t s = rank; s < n; s += world) {
    const uint32_t a = s * R, b = (a + R < height) ? a + R : height;
    v.push_back(StripeSpan{a, b - a, at});
    at += b - a;
  }
  return v;
}
// packed[rank r][chunkRows][rowBytes] (rank-major, as an all-gather leaves it) -> frame[height][rowBytes]
inline void unpackStripes(const uint8_t* packed, uint8_t* frame, uint32_t height, uint32_t world, uint32_t R, size_t rowBytes) {
  const size_t perRank = (size_t)stripeChunkRows(height, world, R) * rowBytes;
  for (uint32_t r = 0; r < world; r++)
    for (const StripeSpan& s : stripeSpans(height, world, r, R))
      std::memcpy(frame + (size_t)s.firstRow * rowBytes, packed + (size_t)r * perRank + (size_t)s.packedRow * rowBytes, (size_t)s.rows * rowBytes);
}

// ---- id file ------------------------------------------------------------------------------------------------------
// [magic "BDPTNCID"][nonce u64][payload bytes].  Rank 0 removes whatever is at `path`, writes under a temporary name
// and renames (readers never see half a file), and removes the file again once every rank has joined (retireIdFile).
// The other ranks accept a file only if its nonce is theirs — the launcher gives every run of a job its own
// (bdpt_render --job-id) — so the id of a crashed earlier run is skipped and they keep waiting for this run's.
constexpr char kIdFileMagic[8] = {'B', 'D', 'P', 'T', 'N', 'C', 'I', 'D'};
inline uint64_t jobNonce(const std::string& jobId) {  // FNV-1a; "" -> 0 (no job id given: only the single-use rule protects)
  if (jobId.empty()) return 0;
  uint64_t h = 1469598103934665603ull;
  for (unsigned char ch : jobId) h = (h ^ ch) * 1099511628211ull;
  return h ? h : 1;
}
inline bool writeIdFile(const std::string& path, uint64_t nonce, const void* payload, size_t bytes) {
  (void)std::remove(path.c_str());  // single use: an earlier run's file must not be what a fast peer finds
  const std::string tmp = path + ".tmp";
  FILE* f = std::fopen(tmp.c_str(), "wb");
  if (!f) return false;
  bool ok = std::fwrite(kIdFileMagic, 1, 8, f) == 8 && std::fwrite(&nonce, 8, 1, f) == 1 && std::fwrite(payload, 1, bytes, f) == bytes;
  ok = (std::fclose(f) == 0) && ok;
  if (!ok) {
    (void)std::remove(tmp.c_str());
    return false;
  }
  return std::rename(tmp.c_str(), path.c_str()) == 0;
}
// one attempt: true when `path` holds a complete file of this job
inline bool readIdFile(const std::string& path, uint64_t nonce, void* payload, size_t bytes) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) return false;
  char magic[8];
  uint64_t n = 0;
  std::vector<uint8_t> body(bytes + 1);
  bool ok = std::fread(magic, 1, 8, f) == 8 && std::fread(&n, 8, 1, f) == 1;
  const size_t got = ok ? std::fread(body.data(), 1, bytes + 1, f) : 0;
  std::fclose(f);
  if (!ok || std::memcmp(magic, kIdFileMagic, 8) != 0 || n != nonce || got != bytes) return false;
  std::memcpy(payload, body.data(), bytes);
  return true;
}
inline bool waitForIdFile(const std::string& path, uint64_t nonce, void* payload, size_t bytes, double timeoutSeconds) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    if (readIdFile(path, nonce, payload, bytes)) return true;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeoutSeconds) return false;
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}
inline void retireIdFile(const std::string& path) { (void)std::remove(path.c_str()); }

// ---- ranks as threads of one process --------------------------------------------------------------------------------
class RankGroup {
 public:
  explicit RankGroup(int n) : mN(n) {}
  // All ranks meet here.  false: the group was aborted (before or while waiting) — the caller leaves.
  bool wait() {
    std::unique_lock<std::mutex> lk(mM);
    if (mAborted) return false;
    const uint64_t gen = mGen;
    if (++mCount == mN) {
      mCount = 0;
      mGen++;
      mCv.notify_all();
      return true;
    }
    mCv.wait(lk, [&] { return gen != mGen || mAborted; });
    return gen != mGen;  // (released by the last arrival, not by the abort)
  }
  // Agreement point: every rank reports; true only when ALL did well.  One failure releases everybody with false.
  bool agree(bool ok) {
    if (!ok) mAllOk.store(false);
    const bool met = wait();
    return met && mAllOk.load();
  }
  // A rank that cannot go on (a failure after the agreement point): peers blocked in wait() / agree() return false;
  // peers blocked in a collective are released by the owner of the communicators (onAbort: ncclCommAbort on each).
  void abort() {
    {
      std::lock_guard<std::mutex> lk(mM);
      if (mAborted) return;
      mAborted = true;
    }
    mCv.notify_all();
  }
  bool aborted() {
    std::lock_guard<std::mutex> lk(mM);
    return mAborted;
  }

 private:
  std::mutex mM;
  std::condition_variable mCv;
  const int mN;
  int mCount = 0;
  uint64_t mGen = 0;
  bool mAborted = false;
  std::atomic<bool> mAllOk{true};
};

}  // namespace bdpt
