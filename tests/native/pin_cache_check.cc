// Host-side check of octopuszk_amd/csrc/pin_cache.h (no HIP): eight threads cycle more keys than the cache holds,
// every "build" and every "free" sleeps — as a hipMalloc / a device-synchronising hipFree would — and
//   * calls back into the cache (stats() takes the cache's lock): a build or a free under the lock would deadlock;
//   * counts how many builds are in flight at once: > 1 proves that builds of different keys overlap;
//   * no object is used after it was freed, used before it was published, or freed twice;
//   * the item count and the byte budget hold whenever nothing is pinned.
// Returns 0 when everything held; a bit mask of what failed otherwise.  figures[0..5] = builds, frees, waits on a
// builder, maximum builds in flight at once, items left, bytes left.
#include <atomic>
#include <chrono>
#include <cstdint>
#include <thread>
#include <vector>

#include "../../octopuszk_amd/csrc/pin_cache.h"

using namespace ozk;

namespace {
struct Obj : PinCacheItem {
  int key = 0;
  std::atomic<int> alive{0};   // 1 between build and free
  std::atomic<int> ready{0};
  bool same_key(const Obj& o) const { return key == o.key; }
};
PinCache<Obj> cache;
std::atomic<int> builds{0}, frees{0}, in_build{0}, max_in_build{0}, errors{0};

void free_dead(std::vector<Obj*>& dead) {
  for (Obj* o : dead) {
    std::this_thread::sleep_for(std::chrono::microseconds(300));   // a hipFree
    int it, dummy;
    size_t by;
    unsigned long long b, w;
    cache.stats(0, &it, &by, &b, &w);   // would deadlock under the cache's lock
    (void)dummy;
    if (o->state != PIN_SEEN && o->state != PIN_FAILED && o->alive.exchange(0) != 1) errors |= 1;   // double free / never built
    if (o->refs != 0) errors |= 2;
    frees++;
    delete o;
  }
  dead.clear();
}
}  // namespace

extern "C" int pin_cache_check(int threads, int keys, int iters, int second_use, int fail_every, long long* figures) {
  const PinCacheLimits lim{4, (size_t)4 * 100};
  std::vector<std::thread> th;
  for (int t = 0; t < threads; t++)
    th.emplace_back([=] {
      uint32_t rng = 12345u + 977u * (uint32_t)t;
      for (int i = 0; i < iters; i++) {
        rng = rng * 1664525u + 1013904223u;
        Obj key;
        key.device = 0;
        key.key = (int)((rng >> 8) % (uint32_t)keys);
        std::vector<Obj*> dead;
        Obj* o = nullptr;
        const auto res = cache.acquire(
            key, 100, lim, second_use != 0,
            [&]() -> Obj* {
              Obj* n = new Obj();
              n->device = 0;
              n->key = key.key;
              return n;
            },
            &o, &dead);
        free_dead(dead);
        if (res == PinCache<Obj>::PER_CALL || res == PinCache<Obj>::BUILD_FAILED) continue;
        if (res == PinCache<Obj>::BUILD) {
          const int nb = ++in_build;
          int m = max_in_build.load();
          while (nb > m && !max_in_build.compare_exchange_weak(m, nb)) {
          }
          std::this_thread::sleep_for(std::chrono::milliseconds(2));   // a hipMalloc + the build enqueue
          int it;
          size_t by;
          unsigned long long b, w;
          cache.stats(0, &it, &by, &b, &w);   // would deadlock under the cache's lock
          const bool ok = !(fail_every > 0 && (builds.load() % fail_every) == fail_every - 1);
          builds++;
          if (ok) {
            o->alive = 1;
            o->ready = 1;
          }
          --in_build;
          cache.publish(o, ok, &dead);
          free_dead(dead);
          if (!ok) continue;
        }
        // use it while pinned
        if (o->key != key.key) errors |= 4;
        if (o->ready.load() != 1 || o->alive.load() != 1) errors |= 8;   // used before published / after freed
        std::this_thread::sleep_for(std::chrono::microseconds(100));
        if (o->alive.load() != 1) errors |= 16;   // freed while pinned
        cache.release(o, lim, &dead);
        free_dead(dead);
      }
    });
  for (auto& t : th) t.join();
  int items;
  size_t bytes;
  unsigned long long b, w;
  cache.stats(0, &items, &bytes, &b, &w);
  if (items > lim.max_items || bytes > lim.max_bytes) errors |= 32;
  figures[0] = builds;
  figures[1] = frees;
  figures[2] = (long long)w;
  figures[3] = max_in_build;
  figures[4] = items;
  figures[5] = (long long)bytes;
  std::vector<Obj*> dead;
  cache.drain(&dead);
  free_dead(dead);
  cache.stats(0, &items, &bytes, &b, &w);
  if (items != 0) errors |= 64;
  return errors.load();
}
