// partition.h -- how the row-partitioned `hip` target cuts a matrix over its ranks (SURVEY
// 8e; no counterpart in the single-process reference).  Pure host arithmetic, no GPU, no
// communication: every rank computes the same cut from the same input, and the tests check
// it on the CPU (tests/test_partition.py through libabft_host.so).
//
//   * contiguous blocks of output indices (CSR: rows, COO: columns) with (nearly) equal
//     non-zero counts, at least one output each;
//   * every vector is a slice inside a "gathered" buffer of size() equal slots (slot = the
//     longest block), so any vector can be an SpMV input after one exchange; gather indices
//     (CSR: columns, COO: rows) are re-based to that padded layout;
//   * per peer, the window of its slot this rank reads (banded matrices exchange only these
//     halos instead of an all-gather), and the longest run of local outputs that read
//     nothing from a peer (they are multiplied while the exchange is in flight).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

struct ShardPlan
{
  int ranks, me, N;
  std::vector<int> bounds;        // ranks + 1 output-index boundaries
  int slot, n_pad, out0, n_loc;   // slot length, padded vector length, this rank's outputs [out0, out0 + n_loc)
  // this rank's elements, in the caller's order, indices re-based (see above)
  std::vector<uint32_t> lout, pin;   // local output index, padded gather index
  std::vector<double> vals;
  // caller's (global) element index of local element k: first + k when contiguous (CSR),
  // else listed (COO: a column block's elements are scattered over the row-major input)
  size_t first;
  std::vector<uint32_t> global_index;  // empty when contiguous
  std::vector<int> need;          // 2 * ranks: window [lo, hi) of each peer's slot this rank reads
  int interior_lo, interior_hi;   // longest run of local outputs reading only the own slot ([0,0): none)
};

// equal-nnz cut of outputs [0, N) given the per-element output index array sorted ascending
// (CSR rows).  Every rank gets at least one output; needs N >= ranks.
void plan_bounds_sorted(const uint32_t *out_sorted, long long nnz, int N, int ranks, std::vector<int> &bounds);
// same cut from per-output counts (COO: elements per column)
void plan_bounds_counts(const std::vector<long long> &per_output, int N, int ranks, std::vector<int> &bounds);

// fmt 0 = CSR (output = rows[], gather = columns[]), 1 = COO (output = columns[], gather = rows[]).
// `columns/rows/values`: the elements [elem0, elem0 + count) of the caller's arrays that may
// belong to this rank (the whole matrix, or -- CSR, generated block-wise -- just its block);
// `bounds` as computed above.
void plan_shard(int fmt, const uint32_t *columns, const uint32_t *rows, const double *values, size_t elem0,
                size_t count, int N, const std::vector<int> &bounds, int me, ShardPlan &plan);

// The halo windows as they travel through shared host memory (include/abft_hip.h,
// abft_hip_peer_exchange_*): every rank lays the windows its peers read out in an outbox -- in
// ascending reader order, each followed by the library's 8-byte check word, 256-byte aligned --
// and since `all_need` (ranks * ranks windows: [2 * (q * ranks + g)] = what rank q reads of rank
// g's slot) is known to every rank, all of them compute the same layout without talking.
struct WindowPiece
{
  int peer;              // out: the reader; in: the sender
  uint32_t vector_offset, count;   // doubles from the start of the gathered vector
  uint64_t box_offset;   // bytes from the start of the SENDER's outbox
};
// returns the outbox size every rank uses (the largest rank's; 0: no windows at all)
size_t plan_outboxes(const std::vector<int> &all_need, int ranks, int slot, int me, std::vector<WindowPiece> &out,
                     std::vector<WindowPiece> &in);

extern "C" {
// ctypes view of plan_outboxes: pieces as 4 x int64 {peer, vector_offset, count, box_offset}; returns the
// outbox size, *nout / *nin the piece counts (at most `cap` of each are written)
long long abft_plan_outboxes(const int *all_need, int ranks, int slot, int me, long long *out, int *nout, long long *in,
                             int *nin, int cap);
// ctypes view for the CPU tests: returns 0, fills bounds[ranks+1], scalars[8] = {slot, n_pad, out0,
// n_loc, local nnz, first, interior_lo, interior_hi}, need[2*ranks]; lout/pin/gidx (each `cap`
// entries, may be NULL) receive the local element arrays.
int abft_plan_shard(int fmt, const uint32_t *columns, const uint32_t *rows, const double *values, long long nnz,
                    int N, int ranks, int me, int *bounds, long long *scalars, int *need, uint32_t *lout,
                    uint32_t *pin, uint32_t *gidx, long long cap);
}
