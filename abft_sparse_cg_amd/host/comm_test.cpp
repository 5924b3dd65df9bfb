// comm_test -- exercises comm.h's host collectives between WORLD_SIZE processes
// (tests/test_host_logic.py starts them); prints "ok" on every rank.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>

#include "comm.h"

#define REQUIRE(c) do { if (!(c)) { fprintf(stderr, "rank %d: %s failed (line %d)\n", r, #c, __LINE__); return 1; } } while (0)

int main()
{
  Comm *c = Comm::from_env();
  if (!c) { printf("single\n"); return 0; }
  const int r = c->rank(), n = c->size();
  // (test aid: rank 0 leaves with status 2 right after the rendezvous while the others sit in a receive
  // from it -- the exit handler of a self-launching rank 0 must end them, not wait for them for ever)
  if (getenv("COMM_TEST_RANK0_LEAVES") && r == 0)
  {
    usleep(300 * 1000);
    fprintf(stderr, "rank 0: leaving with status 2 as asked\n");
    exit(2);
  }
  // bcast from every root
  for (int root = 0; root < n; root++)
  {
    int v[3] = {r == root ? 100 + root : -1, r == root ? 7 : -1, r == root ? root * root : -1};
    c->bcast(v, sizeof(v), root);
    REQUIRE(v[0] == 100 + root && v[1] == 7 && v[2] == root * root);
  }
  // allreduce: rank-ordered sum, identical bits everywhere
  double s[2] = {1.0 / (r + 1), (double)r};
  c->allreduce_sum(s, 2);
  double want = 0.0;
  for (int k = 0; k < n; k++) want += 1.0 / (k + 1);
  REQUIRE(s[0] == want && s[1] == n * (n - 1) / 2.0);
  // allgather (equal sizes)
  std::vector<int> all(n);
  int mine = r * 3 + 1;
  c->allgather(&mine, sizeof(mine), all.data());
  for (int k = 0; k < n; k++) REQUIRE(all[k] == k * 3 + 1);
  // allgatherv: rank k contributes k doubles (rank 0 nothing)
  std::vector<double> part(r);
  for (int i = 0; i < r; i++) part[i] = r + 0.25 * i;
  std::vector<char> bytes;
  std::vector<size_t> sizes;
  c->allgatherv(part.data(), part.size() * sizeof(double), bytes, sizes);
  REQUIRE((int)sizes.size() == n);
  size_t off = 0;
  for (int k = 0; k < n; k++)
  {
    REQUIRE(sizes[k] == k * sizeof(double));
    for (int i = 0; i < k; i++)
    {
      double d;
      memcpy(&d, bytes.data() + off + i * sizeof(double), sizeof(d));
      REQUIRE(d == k + 0.25 * i);
    }
    off += sizes[k];
  }
  REQUIRE(bytes.size() == off);
  // a large message (crosses socket buffer sizes)
  std::vector<double> big(1 << 20, (double)r);
  std::vector<double> got((size_t)n << 20);
  c->allgather(big.data(), big.size() * sizeof(double), got.data());
  for (int k = 0; k < n; k++) REQUIRE(got[((size_t)k << 20) + 12345] == (double)k);
  // point-to-point pieces relayed by rank 0: rank s sends (d + 1) doubles of value 100 s + d to
  // every rank d > s + 0 with (s + d) odd; pieces listed in ascending peer order on both sides
  {
    std::vector<std::vector<double> > obuf(n), ibuf(n);
    std::vector<Comm::Piece> out, in;
    for (int d = 0; d < n; d++)
    {
      if (d == r || ((r + d) & 1) == 0) continue;
      obuf[d].assign(d + 1, 100.0 * r + d);
      out.push_back(Comm::Piece{d, obuf[d].data(), obuf[d].size() * sizeof(double)});
      ibuf[d].assign(r + 1, -1.0);
      in.push_back(Comm::Piece{d, ibuf[d].data(), ibuf[d].size() * sizeof(double)});
    }
    c->exchange(out, in);
    for (int s2 = 0; s2 < n; s2++)
    {
      if (s2 == r || ((r + s2) & 1) == 0) continue;
      for (int i = 0; i <= r; i++) REQUIRE(ibuf[s2][i] == 100.0 * s2 + r);
    }
  }
  REQUIRE(!c->device_collectives());
  c->barrier();
  delete c;
  // (test aid: the rank named here ends with status 3 after a clean run -- what a parent that started its
  // ranks itself, ABFT_HIP_GPUS, must not hide behind its own 0)
  const char *bad = getenv("COMM_TEST_FAIL_RANK");
  if (bad && atoi(bad) == r)
  {
    fprintf(stderr, "rank %d: leaving with status 3 as asked\n", r);
    return 3;
  }
  printf("ok\n");
  return 0;
}
