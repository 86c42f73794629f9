// host_logic_test.cpp — the parts of the multi-GPU host that need no GPU (host/RankSync.h), run on the CPU by
// tests/test_cpu_oracle_and_host.py::test_host_rank_logic_*:
//   stripes <H> <world>            prints the stripe spans of every rank and the chunk rows (compared with tiling.py), and
//                                  checks that pack -> all-gather layout -> unpackStripes is the identity on a frame
//   idfile <dir>                   the single-use, nonce-carrying id file: a stale file is not accepted
//   ranks                          RankGroup: agreement with one failing rank, abort releasing ranks blocked in wait()
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "RankSync.h"

using namespace bdpt;

static int fail(const char* what) {
  std::printf("FAIL %s\n", what);
  return 1;
}

static int stripes(uint32_t H, uint32_t world, uint32_t R) {
  const uint32_t W = 5, chunk = stripeChunkRows(H, world, R);
  const size_t rowBytes = W * 4 * sizeof(float);
  std::printf("{\"chunk_rows\": %u, \"ranks\": [", chunk);
  std::vector<uint8_t> frame((size_t)H * rowBytes), packed((size_t)world * chunk * rowBytes, 0xEE), back((size_t)H * rowBytes, 0);
  for (size_t i = 0; i < frame.size(); i++) frame[i] = (uint8_t)((i * 2654435761u) >> 13);
  std::vector<int> owner(H, -1);
  for (uint32_t r = 0; r < world; r++) {
    std::printf("%s[", r ? ", " : "");
    uint32_t rows = 0;
    bool first = true;
    for (const StripeSpan& s : stripeSpans(H, world, r, R)) {
      std::printf("%s[%u, %u]", first ? "" : ", ", s.firstRow, s.firstRow + s.rows);
      first = false;
      if (s.packedRow != rows) return fail("packed rows are not consecutive");
      rows += s.rows;
      if (rows > chunk) return fail("a rank's rows exceed the chunk");
      for (uint32_t y = s.firstRow; y < s.firstRow + s.rows; y++) {
        if (y >= H || owner[y] != -1) return fail("a row is dealt twice or lies outside the frame");
        owner[y] = (int)r;
      }
      // what a rank does before the all-gather: its rows, in order, at the head of its chunk
      std::memcpy(&packed[((size_t)r * chunk + s.packedRow) * rowBytes], &frame[(size_t)s.firstRow * rowBytes], (size_t)s.rows * rowBytes);
    }
    std::printf("]");
  }
  std::printf("]}\n");
  for (uint32_t y = 0; y < H; y++)
    if (owner[y] < 0) return fail("a row has no owner");
  unpackStripes(packed.data(), back.data(), H, world, R, rowBytes);
  if (back != frame) return fail("unpackStripes(pack(frame)) != frame");
  return 0;
}

static int idfile(const std::string& dir) {
  const std::string path = dir + "/nccl.id";
  uint8_t id[128], got[128];
  for (int i = 0; i < 128; i++) id[i] = (uint8_t)(i * 7 + 1);
  // a crashed earlier run left its file behind (another nonce, or no nonce at all, or a torn file)
  uint8_t old[128];
  std::memset(old, 0x55, sizeof(old));
  if (!writeIdFile(path, jobNonce("run-1"), old, sizeof(old))) return fail("writeIdFile");
  if (readIdFile(path, jobNonce("run-2"), got, sizeof(got))) return fail("a stale file of another run was accepted");
  if (waitForIdFile(path, jobNonce("run-2"), got, sizeof(got), 0.2)) return fail("waitForIdFile accepted a stale file");
  {
    FILE* f = std::fopen(path.c_str(), "wb");  // the pre-round-5 format: the bare id
    std::fwrite(old, 1, sizeof(old), f);
    std::fclose(f);
  }
  if (readIdFile(path, 0, got, sizeof(got))) return fail("a file without the header was accepted");
  {
    FILE* f = std::fopen(path.c_str(), "wb");  // torn: header of THIS run, half a payload
    const uint64_t n = jobNonce("run-2");
    std::fwrite(kIdFileMagic, 1, 8, f);
    std::fwrite(&n, 8, 1, f);
    std::fwrite(id, 1, 64, f);
    std::fclose(f);
  }
  if (readIdFile(path, jobNonce("run-2"), got, sizeof(got))) return fail("a torn file was accepted");
  // rank 0 of run-2 starts late: a waiting peer takes ITS file, not what was there before
  std::thread rank0([&] {
    std::this_thread::sleep_for(std::chrono::milliseconds(150));
    writeIdFile(path, jobNonce("run-2"), id, sizeof(id));
  });
  const bool ok = waitForIdFile(path, jobNonce("run-2"), got, sizeof(got), 5.0);
  rank0.join();
  if (!ok || std::memcmp(got, id, sizeof(id)) != 0) return fail("the peer did not get this run's id");
  retireIdFile(path);
  if (std::FILE* f = std::fopen(path.c_str(), "rb")) {
    std::fclose(f);
    return fail("the id file survives retireIdFile");
  }
  if (jobNonce("") != 0 || jobNonce("a") == jobNonce("b") || jobNonce("a") == 0) return fail("jobNonce");
  std::printf("idfile ok\n");
  return 0;
}

static int ranks() {
  // (1) set-up fails on rank 2 of 4: every rank learns it at the agreement point and leaves; nobody waits for ever
  {
    RankGroup g(4);
    std::vector<int> left(4, 0);
    std::vector<std::thread> t;
    for (int r = 0; r < 4; r++)
      t.emplace_back([&, r] {
        if (r == 1) std::this_thread::sleep_for(std::chrono::milliseconds(50));  // ranks arrive at different times
        const bool all = g.agree(r != 2);
        if (!all) {
          left[(size_t)r] = 1;
          return;
        }
        g.wait();  // (would be the timing barrier: must not be reached)
        left[(size_t)r] = 2;
      });
    for (auto& th : t) th.join();
    for (int v : left)
      if (v != 1) return fail("agree(false) on one rank did not make every rank leave");
  }
  // (2) all well: agreement holds, barriers work repeatedly
  {
    RankGroup g(3);
    std::vector<int> rounds(3, 0);
    std::vector<std::thread> t;
    for (int r = 0; r < 3; r++)
      t.emplace_back([&, r] {
        if (!g.agree(true)) return;
        for (int k = 0; k < 5; k++)
          if (g.wait()) rounds[(size_t)r]++;
      });
    for (auto& th : t) th.join();
    for (int v : rounds)
      if (v != 5) return fail("barrier rounds");
  }
  // (3) a rank fails AFTER the agreement (its read-back): it aborts, peers blocked in the next barrier come back with
  // false, and a rank that arrives later is not held either
  {
    RankGroup g(3);
    std::vector<int> out(3, -1);
    std::vector<std::thread> t;
    for (int r = 0; r < 3; r++)
      t.emplace_back([&, r] {
        if (!g.agree(true)) return;
        if (r == 0) {
          std::this_thread::sleep_for(std::chrono::milliseconds(80));
          g.abort();
          out[0] = 0;
          return;
        }
        if (r == 2) std::this_thread::sleep_for(std::chrono::milliseconds(200));  // arrives after the abort
        out[(size_t)r] = g.wait() ? 1 : 0;
      });
    for (auto& th : t) th.join();
    if (out[0] != 0 || out[1] != 0 || out[2] != 0 || !g.aborted()) return fail("abort did not release the waiting ranks");
  }
  std::printf("ranks ok\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc >= 4 && std::strcmp(argv[1], "stripes") == 0) {
    const uint32_t H = (uint32_t)std::atoi(argv[2]), world = (uint32_t)std::atoi(argv[3]);
    const uint32_t R = argc >= 5 ? (uint32_t)std::atoi(argv[4]) : 1;
    return stripes(H, world, R);
  }
  if (argc >= 3 && std::strcmp(argv[1], "idfile") == 0) return idfile(argv[2]);
  if (argc >= 2 && std::strcmp(argv[1], "ranks") == 0) return ranks();
  std::fprintf(stderr, "usage: host_logic_test stripes H WORLD R | idfile DIR | ranks\n");
  return 2;
}
