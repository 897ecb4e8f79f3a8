// Temporal object matching (index-only, no gradients), gfx950.
//
// Replaces the T-serial loops of Stove._3_only_match_objects (model/video_prediction/stove.py:200-329)
// and Stove._greedy_match_objects (stove.py:432-514): ~15 tiny ATen launches and one host
// synchronisation (`if num_faults > 0`) per frame in the reference, one launch here.
// Sequences are independent, so one lane walks one sequence through time and emits the
// permutation idx[b][t][a] = index of the current object assigned to slot a; the caller
// applies it with a single differentiable gather.
//   mode 0 ('3_only'): every slot takes its nearest current object; if that is not a
//          permutation, slots are assigned greedily in slot order with column knock-out.
//   mode 1 ('greedy'): N rounds of global arg-min over the N x N distance table with row and
//          column knock-out.
//   mode 2 ('volatile'): nearest current object per slot, no uniqueness, no repair (stove.py:331-430).
// Distances are squared Euclidean on (v+1)/2-scaled features, ties resolve to the first index.
#include "common.h"

namespace stove {

constexpr int kMatchN = 8, kMatchF = 8;

// One wave per sequence: the 64 lanes stage the sequence's whole feature track into LDS with
// coalesced loads (removing T dependent global-load latencies), lane 0 then walks it; the
// permutations are buffered in LDS and written back coalesced.
template <int TN, int TF>
__global__ __launch_bounds__(64) void match_objects_k(const float* __restrict__ feat, long long* __restrict__ idx_out,
                                                      float* __restrict__ perm_out, int B, int T, int Nrt, int Frt, int mode) {
  // compile-time object / feature counts keep every table in registers (fully unrolled loops)
  const int N = TN > 0 ? TN : Nrt;
  const int F = TF > 0 ? TF : Frt;
  extern __shared__ float mlds[];                 // [T*N*F] features, then [T*N] int indices
  const int b = blockIdx.x;
  const int n_feat = T * N * F;
  int* ibuf = reinterpret_cast<int*>(mlds + n_feat);
  for (int i = threadIdx.x; i < n_feat; i += 64) mlds[i] = feat[(size_t)b * n_feat + i];
  __syncthreads();
  if (threadIdx.x == 0) {
  constexpr int MN = TN > 0 ? TN : kMatchN, MF = TF > 0 ? TF : kMatchF;
  float prev[MN][MF], cur[MN][MF], err[MN][MN];
  const float* fb = mlds;
  _Pragma("unroll") for (int a = 0; a < N; ++a) {
    _Pragma("unroll") for (int f = 0; f < F; ++f) prev[a][f] = (fb[a * F + f] + 1.0f) * 0.5f;
    ibuf[a] = a;
  }
  for (int t = 1; t < T; ++t) {
    _Pragma("unroll") for (int j = 0; j < N; ++j)
      _Pragma("unroll") for (int f = 0; f < F; ++f) cur[j][f] = (fb[((size_t)t * N + j) * F + f] + 1.0f) * 0.5f;
    // err[a][j] = | prev_a - cur_j |^2
    _Pragma("unroll") for (int a = 0; a < N; ++a)
      _Pragma("unroll") for (int j = 0; j < N; ++j) {
        float s = 0.0f;
        _Pragma("unroll") for (int f = 0; f < F; ++f) {
          const float d = prev[a][f] - cur[j][f];
          s += d * d;
        }
        err[a][j] = s;
      }
    int idx[MN];
    if (mode == 0) {
      _Pragma("unroll") for (int a = 0; a < N; ++a) {
        int best = 0;
        _Pragma("unroll") for (int j = 1; j < N; ++j)
          if (err[a][j] < err[a][best]) best = j;
        idx[a] = best;
      }
      bool ok = true;
      _Pragma("unroll") for (int a = 0; a < N; ++a)
        _Pragma("unroll") for (int c = a + 1; c < N; ++c)
          if (idx[a] == idx[c]) ok = false;
      if (!ok) {
        _Pragma("unroll") for (int a = 0; a < N; ++a) {
          int best = 0;
          _Pragma("unroll") for (int j = 1; j < N; ++j)
            if (err[a][j] < err[a][best]) best = j;
          idx[a] = best;
          _Pragma("unroll") for (int r = 0; r < N; ++r) err[r][best] = 1e12f;
        }
      }
    } else if (mode == 1) {
      _Pragma("unroll") for (int a = 0; a < N; ++a) idx[a] = 0;
      _Pragma("unroll") for (int round = 0; round < N; ++round) {
        int ba = 0, bj = 0;
        float bv = err[0][0];
        _Pragma("unroll") for (int a = 0; a < N; ++a)
          _Pragma("unroll") for (int j = 0; j < N; ++j)
            if (err[a][j] < bv) {
              bv = err[a][j];
              ba = a;
              bj = j;
            }
        idx[ba] = bj;
        _Pragma("unroll") for (int q = 0; q < N; ++q) {
          err[ba][q] = 3.0e38f;
          err[q][bj] = 3.0e38f;
        }
      }
    } else {
      // volatile (stove.py:331-430): every slot takes its nearest current object, duplicates allowed
      // (errors.min(-2) there runs over the CURRENT objects, so it is mode 0 without the repair)
      _Pragma("unroll") for (int a = 0; a < N; ++a) {
        int best = 0;
        _Pragma("unroll") for (int j = 1; j < N; ++j)
          if (err[a][j] < err[a][best]) best = j;
        idx[a] = best;
      }
    }
    _Pragma("unroll") for (int a = 0; a < N; ++a) {
      ibuf[t * N + a] = idx[a];
      _Pragma("unroll") for (int f = 0; f < F; ++f) prev[a][f] = cur[idx[a]][f];
    }
  }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < T * N; i += 64) idx_out[(size_t)b * T * N + i] = ibuf[i];
}

}  // namespace stove
