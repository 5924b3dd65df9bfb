// partition.cpp -- see partition.h.
#include "partition.h"

#include <algorithm>

void plan_bounds_sorted(const uint32_t *out_sorted, long long nnz, int N, int ranks, std::vector<int> &bounds)
{
  bounds.assign((size_t)ranks + 1, 0);
  bounds[ranks] = N;
  for (int g = 1; g < ranks; g++)
  {
    int b = nnz ? (int)out_sorted[(size_t)((unsigned long long)nnz * g / ranks)] : (int)((long long)N * g / ranks);
    b = std::max(b, bounds[g - 1] + 1);   // at least one output each ...
    b = std::min(b, N - (ranks - g));     // ... and room for the ranks behind
    bounds[g] = b;
  }
}

void plan_bounds_counts(const std::vector<long long> &per_output, int N, int ranks, std::vector<int> &bounds)
{
  long long nnz = 0;
  for (int i = 0; i < N; i++) nnz += per_output[i];
  bounds.assign((size_t)ranks + 1, 0);
  bounds[ranks] = N;
  long long run = 0;
  int o = 0;
  for (int g = 1; g < ranks; g++)
  {
    // the output that holds element number nnz * g / ranks (what the sorted form reads off directly)
    const long long target = (long long)((unsigned long long)nnz * g / ranks);
    while (o < N && run + per_output[o] <= target) run += per_output[o++];
    int b = nnz ? o : (int)((long long)N * g / ranks);
    b = std::max(b, bounds[g - 1] + 1);
    b = std::min(b, N - (ranks - g));
    bounds[g] = b;
  }
}

void plan_shard(int fmt, const uint32_t *columns, const uint32_t *rows, const double *values, size_t elem0,
                size_t count, int N, const std::vector<int> &bounds, int me, ShardPlan &p)
{
  const int G = (int)bounds.size() - 1;
  p.ranks = G; p.me = me; p.N = N; p.bounds = bounds;
  p.slot = 0;
  for (int g = 0; g < G; g++) p.slot = std::max(p.slot, bounds[g + 1] - bounds[g]);
  p.n_pad = p.slot * G;
  p.out0 = bounds[me];
  p.n_loc = bounds[me + 1] - p.out0;
  const uint32_t *out = fmt == 0 ? rows : columns, *in = fmt == 0 ? columns : rows;
  const uint32_t lo = (uint32_t)p.out0, hi = (uint32_t)(p.out0 + p.n_loc);
  p.lout.clear(); p.pin.clear(); p.vals.clear(); p.global_index.clear();
  p.first = 0;
  size_t k0 = 0, k1 = count;
  if (fmt == 0)
  {
    // rows are sorted: this rank's elements are one contiguous run
    k0 = std::lower_bound(out, out + count, lo) - out;
    k1 = std::lower_bound(out, out + count, hi) - out;
    p.first = elem0 + k0;
  }
  p.need.assign(2 * (size_t)G, 0);
  for (int g = 0; g < G; g++) p.need[2 * g] = p.slot;
  std::vector<char> remote((size_t)p.n_loc, 0);
  for (size_t k = k0; k < k1; k++)
  {
    if (out[k] < lo || out[k] >= hi) continue;  // COO: not this rank's column
    const uint32_t o = out[k] - lo, c = in[k];
    uint32_t padded = (uint32_t)p.n_pad;  // out of range stays out of range (the kernel reads 0.0, as on one GPU)
    if (c < (uint32_t)N)
    {
      const int owner = (int)(std::upper_bound(bounds.begin(), bounds.end(), (int)c) - bounds.begin()) - 1;
      const int off = (int)c - bounds[owner];
      padded = (uint32_t)(owner * p.slot + off);
      if (owner != me)
      {
        p.need[2 * owner] = std::min(p.need[2 * owner], off);
        p.need[2 * owner + 1] = std::max(p.need[2 * owner + 1], off + 1);
        remote[o] = 1;
      }
    }
    p.lout.push_back(o);
    p.pin.push_back(padded);
    p.vals.push_back(values[k]);
    if (fmt != 0) p.global_index.push_back((uint32_t)(elem0 + k));
  }
  for (int g = 0; g < G; g++)
    if (p.need[2 * g + 1] <= p.need[2 * g]) p.need[2 * g] = p.need[2 * g + 1] = 0;
  int best_lo = 0, best_hi = 0, run = 0;
  for (int r = 0; r <= p.n_loc; r++)
  {
    if (r < p.n_loc && !remote[r]) { run++; continue; }
    if (run > best_hi - best_lo) { best_lo = r - run; best_hi = r; }
    run = 0;
  }
  const bool interior = best_hi - best_lo >= std::max(p.n_loc / 4, 1);
  p.interior_lo = interior ? best_lo : 0;
  p.interior_hi = interior ? best_hi : 0;
}

size_t plan_outboxes(const std::vector<int> &all_need, int ranks, int slot, int me, std::vector<WindowPiece> &out,
                     std::vector<WindowPiece> &in)
{
  out.clear();
  in.clear();
  size_t box = 0;
  for (int g = 0; g < ranks; g++)
  {
    size_t off = 0;
    for (int q = 0; q < ranks; q++)
    {
      if (q == g) continue;
      const int *w = &all_need[2 * ((size_t)q * ranks + g)];  // what rank q reads of rank g's slot
      if (w[1] <= w[0]) continue;
      WindowPiece pc;
      pc.count = (uint32_t)(w[1] - w[0]);
      pc.vector_offset = (uint32_t)((size_t)g * slot + w[0]);
      pc.box_offset = off;
      if (g == me) { pc.peer = q; out.push_back(pc); }
      if (q == me) { pc.peer = g; in.push_back(pc); }
      off += (((size_t)pc.count + 1) * sizeof(double) + 255) & ~(size_t)255;  // (+ the library's check word)
    }
    box = std::max(box, off);
  }
  return box;
}

extern "C" long long abft_plan_outboxes(const int *all_need, int ranks, int slot, int me, long long *out, int *nout,
                                        long long *in, int *nin, int cap)
{
  if (ranks < 1 || me < 0 || me >= ranks || slot < 0 || !all_need) return -1;
  std::vector<int> need(all_need, all_need + 2 * (size_t)ranks * ranks);
  std::vector<WindowPiece> o, i;
  const size_t box = plan_outboxes(need, ranks, slot, me, o, i);
  for (int pass = 0; pass < 2; pass++)
  {
    const std::vector<WindowPiece> &v = pass ? i : o;
    long long *dst = pass ? in : out;
    for (size_t k = 0; k < v.size() && (int)k < cap && dst; k++)
    {
      dst[4 * k] = v[k].peer; dst[4 * k + 1] = v[k].vector_offset; dst[4 * k + 2] = v[k].count;
      dst[4 * k + 3] = (long long)v[k].box_offset;
    }
  }
  if (nout) *nout = (int)o.size();
  if (nin) *nin = (int)i.size();
  return (long long)box;
}

extern "C" int abft_plan_shard(int fmt, const uint32_t *columns, const uint32_t *rows, const double *values,
                               long long nnz, int N, int ranks, int me, int *bounds, long long *scalars, int *need,
                               uint32_t *lout, uint32_t *pin, uint32_t *gidx, long long cap)
{
  if (ranks < 1 || me < 0 || me >= ranks || N < ranks || nnz < 0) return 1;
  std::vector<int> b;
  if (fmt == 0)
    plan_bounds_sorted(rows, nnz, N, ranks, b);
  else
  {
    std::vector<long long> per((size_t)N, 0);
    for (long long i = 0; i < nnz; i++)
      if (columns[i] < (uint32_t)N) per[columns[i]]++;
    plan_bounds_counts(per, N, ranks, b);
  }
  ShardPlan p;
  plan_shard(fmt, columns, rows, values, 0, (size_t)nnz, N, b, me, p);
  for (int g = 0; g <= ranks; g++) bounds[g] = b[g];
  scalars[0] = p.slot; scalars[1] = p.n_pad; scalars[2] = p.out0; scalars[3] = p.n_loc;
  scalars[4] = (long long)p.lout.size(); scalars[5] = (long long)p.first;
  scalars[6] = p.interior_lo; scalars[7] = p.interior_hi;
  for (int g = 0; g < 2 * ranks; g++) need[g] = p.need[g];
  const long long n = std::min<long long>(cap, (long long)p.lout.size());
  for (long long k = 0; k < n; k++)
  {
    if (lout) lout[k] = p.lout[k];
    if (pin) pin[k] = p.pin[k];
    if (gidx) gidx[k] = p.global_index.empty() ? (uint32_t)(p.first + k) : p.global_index[k];
  }
  return 0;
}
