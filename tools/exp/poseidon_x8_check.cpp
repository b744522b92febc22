// ragged batches through the eight-lane sponge engine against the scalar sponge; then timing
#include <chrono>
#include <cstdio>
#include <random>
#include "poseidon.hpp"
using namespace zkhost;
using namespace zkhost::pos;
int main() {
  std::mt19937_64 rng(7);
  auto rnd = [&] { U256 v; for (int i = 0; i < 4; ++i) v.l[i] = rng(); v.l[3] &= 0x0fffffffffffffffULL; return v; };   // below r
  for (int trial = 0; trial < 20; ++trial) {
    const int nj = 1 + rng() % 19;
    std::vector<std::vector<U256>> data(nj);
    std::vector<Sponge> a(nj), b(nj);
    std::vector<AbsorbJob> jobs(nj);
    for (int j = 0; j < nj; ++j) {
      const size_t len = rng() % 70, pre = rng() % 3;
      data[j].resize(len);
      for (auto &v : data[j]) v = rnd();
      for (size_t q = 0; q < pre; ++q) { U256 v = rnd(); a[j].update(v); b[j].update(v); }
      for (auto &v : data[j]) a[j].update(v);
    }
    // bulk path, on this thread: emulate begin_bulk / end_bulk with x8_absorb_now
    std::vector<AbsorbJob *> ptr;
    struct Tail { const U256 *p; };
    for (int j = 0; j < nj; ++j) ptr.push_back(&jobs[j]);
    // use the service (threads) for odd trials, inline for even ones
    if (trial & 1) {
      for (int j = 0; j < nj; ++j) b[j].begin_bulk(data[j].data(), data[j].size(), jobs[j]);
      for (int j = 0; j < nj; ++j) b[j].end_bulk(jobs[j]);
    } else {
      for (int j = 0; j < nj; ++j) b[j].begin_bulk_prepare(data[j].data(), data[j].size(), jobs[j]);
      if (!x8_absorb_now(ptr.data(), ptr.size())) { printf("no IFMA\n"); return 0; }
      for (int j = 0; j < nj; ++j) b[j].end_bulk(jobs[j]);
    }
    for (int j = 0; j < nj; ++j) {
      const U256 x = a[j].squeeze(), y = b[j].squeeze();
      if (memcmp(x.l, y.l, 32)) { printf("MISMATCH trial %d job %d (len %zu)\n", trial, j, data[j].size()); return 1; }
    }
  }
  printf("x8 matches the scalar sponge on 20 ragged batches (service and inline)\n");
  for (int nj : {1, 2, 4, 8, 16, 24}) {
    const size_t len = 5120;
    std::vector<std::vector<U256>> data(nj, std::vector<U256>(len));
    for (auto &d : data) for (auto &v : d) v = rnd();
    std::vector<Sponge> b(nj);
    std::vector<AbsorbJob> jobs(nj);
    auto t0 = std::chrono::steady_clock::now();
    for (int j = 0; j < nj; ++j) b[j].begin_bulk(data[j].data(), len, jobs[j]);
    for (int j = 0; j < nj; ++j) b[j].end_bulk(jobs[j]);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("%2d sponges x %zu values through the service: %.2f ms wall, %.3f us per permutation per sponge\n", nj, len, us / 1e3, us / (len / 2) / nj);
  }
  {
    Sponge s; std::vector<U256> d(5120); for (auto &v : d) v = rnd();
    auto t0 = std::chrono::steady_clock::now();
    for (auto &v : d) s.update(v);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("single sponge (ifma path): %.3f us per permutation\n", us / 2560);
  }
}
