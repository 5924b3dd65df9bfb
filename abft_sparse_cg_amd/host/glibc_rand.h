// glibc_rand.h -- glibc's rand() as a private object: the additive-feedback
// generator r[i] = r[i-3] + r[i-31] (TYPE_3), seeded the way srand(seed) seeds
// it.  The reference fills b from rand() with the default seed 1 before anything
// else touches the generator (cg.cpp:66-74); in a process that has loaded the
// HIP runtime the shared libc state may already have been drawn from, so b is
// produced from an identical private sequence instead (checked against libc in
// tests/test_capi_symbols.py).
#pragma once
#include <cstdint>

class GlibcRand
{
public:
  explicit GlibcRand(unsigned seed = 1)
  {
    int32_t r[34];
    r[0] = seed ? (int32_t)seed : 1;
    for (int i = 1; i < 31; i++)
    {
      int64_t v = (16807LL * r[i - 1]) % 2147483647LL;
      r[i] = (int32_t)(v < 0 ? v + 2147483647LL : v);
    }
    for (int i = 0; i < 31; i++)
      state_[i] = (uint32_t)r[i];
    front_ = 3;
    rear_ = 0;
    for (int i = 0; i < 310; i++)
      next();
  }
  int next()
  {
    state_[front_] += state_[rear_];
    const int out = (int)(state_[front_] >> 1);
    front_ = (front_ + 1) % 31;
    rear_ = (rear_ + 1) % 31;
    return out;
  }
private:
  uint32_t state_[31];
  int front_, rear_;
};
