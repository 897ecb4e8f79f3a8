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

// ---- lane-parallel version -------------------------------------------------------------------------------------
// One wave per sequence as above, but the N x N error table lives across the lanes (lane = a N + j): the errors of a
// frame are one instruction deep, the per-slot arg-mins run on N lanes at once, and the global arg-mins of the greedy
// matcher are two DPP wave reductions per round (value, then lowest lane among the minima: exactly the
// first-in-row-major-order tie rule of the serial loop).  The T-serial walk of lane 0 (60 us for N = 3, 580 us for the
// six-object greedy matcher) was pure latency.  Only the rare repair of '3_only' stays serial.
__device__ __forceinline__ float wave_min_bcast(float v) {
#define STOVE_DPP_MIN(ctrl, rmask) \
  v = fminf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0x7f800000, __builtin_bit_cast(int, v), ctrl, rmask, 0xF, false)))
  STOVE_DPP_MIN(0xB1, 0xF);    // quad_perm [1,0,3,2]
  STOVE_DPP_MIN(0x4E, 0xF);    // quad_perm [2,3,0,1]
  STOVE_DPP_MIN(0x114, 0xF);   // row_shr:4
  STOVE_DPP_MIN(0x118, 0xF);   // row_shr:8
  STOVE_DPP_MIN(0x142, 0xA);   // row_bcast:15
  STOVE_DPP_MIN(0x143, 0xC);   // row_bcast:31
#undef STOVE_DPP_MIN
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

__global__ __launch_bounds__(64) void match_objects_par_k(const float* __restrict__ feat, long long* __restrict__ idx_out, int B, int T,
                                                          int N, int F, int mode) {
  extern __shared__ float mlds[];                 // [T*N*F] features, then [T*N] int indices
  __shared__ float pbuf[kMatchN * kMatchF];       // features of the object currently held by every slot
  __shared__ float ebuf[kMatchN * kMatchN];
  __shared__ int sidx[kMatchN];
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n_feat = T * N * F;
  int* ibuf = reinterpret_cast<int*>(mlds + n_feat);
  for (int i = lane; i < n_feat; i += 64) mlds[i] = feat[(size_t)b * n_feat + i];
  __syncthreads();
  const int a = lane / N, j = lane % N;
  const bool cell = lane < N * N;
  if (lane < N * F) pbuf[lane] = (mlds[lane] + 1.0f) * 0.5f;
  if (lane < N) ibuf[lane] = lane;
  __syncthreads();
  for (int t = 1; t < T; ++t) {
    const float* cur = mlds + (size_t)t * N * F;
    float e = __builtin_inff();
    if (cell) {
      e = 0.0f;
      for (int f = 0; f < F; ++f) {
        const float d = pbuf[a * F + f] - (cur[j * F + f] + 1.0f) * 0.5f;
        e += d * d;
      }
    }
    if (mode == 1) {
      // greedy: N rounds of global arg-min with row and column knock-out
      for (int round = 0; round < N; ++round) {
        const float m = wave_min_bcast(e);
        const int w = (int)wave_min_bcast((cell && e == m) ? (float)lane : 64.0f);
        const int ba = w / N, bj = w % N;
        if (lane == 0) sidx[ba] = bj;
        if (cell && (a == ba || j == bj)) e = 3.0e38f;
      }
    } else {
      if (cell) ebuf[lane] = e;
      __syncthreads();
      if (lane < N) {
        int best = 0;
        for (int q = 1; q < N; ++q)
          if (ebuf[lane * N + q] < ebuf[lane * N + best]) best = q;
        sidx[lane] = best;
      }
      __syncthreads();
      if (mode == 0) {
        bool dup = false;
        if (lane < N)
          for (int c = 0; c < N; ++c)
            if (c != lane && sidx[c] == sidx[lane]) dup = true;
        if (__any(dup)) {
          if (lane == 0) {      // rare: slots in order, each takes its nearest object that is still free
            for (int s = 0; s < N; ++s) {
              int best = 0;
              for (int q = 1; q < N; ++q)
                if (ebuf[s * N + q] < ebuf[s * N + best]) best = q;
              sidx[s] = best;
              for (int r = 0; r < N; ++r) ebuf[r * N + best] = 1e12f;
            }
          }
        }
      }
    }
    __syncthreads();
    if (lane < N) ibuf[t * N + lane] = sidx[lane];
    if (lane < N * F) pbuf[lane] = (cur[sidx[lane / F] * F + lane % F] + 1.0f) * 0.5f;
    __syncthreads();
  }
  for (int i = lane; i < T * N; i += 64) idx_out[(size_t)b * T * N + i] = ibuf[i];
}

// ---- greedy matcher, any N: the walk through time as a composition of per-frame assignments --------------------------------
// The greedy rule (stove.py:432-514: N rounds of global arg-min over the N x N distance table with row / column knock-out) picks its
// pairs by VALUE: which previous object goes with which current one does not depend on the order the previous objects sit in the
// slots -- unless two remaining entries tie exactly for a round's minimum (then the first in slot-major order wins).  So every frame
// computes, on the RAW object order and with no dependence on other frames, its assignment A_t: raw object i of frame t-1 -> raw
// object of frame t, plus a flag "a round had a tie" (match_greedy_frames_k: one LANE per frame, the table in registers); the walk
// (match_greedy_compose_k: one wave per sequence) is idx_t[a] = A_t[idx_{t-1}[a]], one dependent LDS read per frame, and redoes a
// flagged frame with the serial rule in slot order.  Same float operations on the same values as the serial walk: identical indices
// (the tests compare them on tracks full of exact ties).  Six objects, T = 100, 256 sequences: 175 us -> ~15 us.
__device__ __forceinline__ float match_err(const float* __restrict__ p, const float* __restrict__ c, int F) {
  float e = 0.0f;
  for (int f = 0; f < F; ++f) {
    const float d = (p[f] + 1.0f) * 0.5f - (c[f] + 1.0f) * 0.5f;
    e = fmaf(d, d, e);
  }
  return e;
}
// greedy on a table in registers (rows = holders of the previous objects in the given order); returns true if a round's minimum was
// attained more than once.  assign[a] = column of row a.
__device__ __forceinline__ bool greedy_rounds(float (&e)[kMatchN][kMatchN], int N, int* assign) {
  bool tie = false;
  for (int round = 0; round < N; ++round) {
    float bv = __builtin_inff();
    int ba = 0, bj = 0, cnt = 0;
#pragma unroll
    for (int a = 0; a < kMatchN; ++a)
#pragma unroll
      for (int j = 0; j < kMatchN; ++j) {
        if (a < N && j < N) {
          const float v = e[a][j];
          cnt = v < bv ? 1 : (v == bv ? cnt + 1 : cnt);
          if (v < bv) {
            bv = v;
            ba = a;
            bj = j;
          }
        }
      }
    tie = tie || (cnt > 1 && bv < 3.0e38f);
    assign[ba] = bj;
#pragma unroll
    for (int a = 0; a < kMatchN; ++a)
#pragma unroll
      for (int j = 0; j < kMatchN; ++j)
        if (a == ba || j == bj) e[a][j] = 3.0e38f;
  }
  return tie;
}
// scratch layout inside the idx block of sequence b (T N int64): the last T N + T bytes = [A (T x N bytes) | tie flags (T bytes)]
__device__ __forceinline__ unsigned char* greedy_scratch(long long* idx_out, int b, int T, int N) {
  return reinterpret_cast<unsigned char*>(idx_out + (size_t)(b + 1) * T * N) - (size_t)T * N - T;
}
__global__ __launch_bounds__(256) void match_greedy_frames_k(const float* __restrict__ feat, long long* __restrict__ idx_out, int B, int T, int N, int F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, t = i % T;
  unsigned char* sc = greedy_scratch(idx_out, b, T, N);
  if (t == 0) {
    for (int a = 0; a < N; ++a) sc[a] = (unsigned char)a;
    sc[(size_t)T * N] = 0;
    return;
  }
  const float* prev = feat + ((size_t)b * T + t - 1) * N * F;
  const float* cur = prev + (size_t)N * F;
  float e[kMatchN][kMatchN];
#pragma unroll
  for (int a = 0; a < kMatchN; ++a)
#pragma unroll
    for (int j = 0; j < kMatchN; ++j) e[a][j] = (a < N && j < N) ? match_err(prev + a * F, cur + j * F, F) : 3.0e38f;
  int assign[kMatchN];
#pragma unroll
  for (int a = 0; a < kMatchN; ++a) assign[a] = 0;
  const bool tie = greedy_rounds(e, N, assign);
#pragma unroll
  for (int a = 0; a < kMatchN; ++a)
    if (a < N) sc[(size_t)t * N + a] = (unsigned char)assign[a];
  sc[(size_t)T * N + t] = tie ? 1 : 0;
}
__global__ __launch_bounds__(64) void match_greedy_compose_k(const float* __restrict__ feat, long long* __restrict__ idx_out, int B, int T, int N, int F) {
  extern __shared__ unsigned char glds[];          // [T*N] assignments, [T] tie flags, then (4-byte aligned) [T*N] int indices
  const int b = blockIdx.x, lane = threadIdx.x;
  const unsigned char* sc = greedy_scratch(idx_out, b, T, N);
  const int nsc = T * N + T;
  for (int i = lane; i < nsc; i += 64) glds[i] = sc[i];
  int* ibuf = reinterpret_cast<int*>(glds + ((nsc + 3) & ~3));
  __shared__ int slot[kMatchN];
  __syncthreads();
  int r = lane < N ? lane : 0;                     // raw object of the current frame held by slot `lane`
  if (lane < N) ibuf[lane] = r;
  for (int t = 1; t < T; ++t) {
    if (glds[T * N + t] == 0) {                    // wave-uniform
      r = glds[t * N + r];
    } else {
      // a tie somewhere in this frame's table: the reference's rule in SLOT order (lane 0 walks the table)
      if (lane < N) slot[lane] = r;
      __syncthreads();
      if (lane == 0) {
        const float* prev = feat + ((size_t)b * T + t - 1) * N * F;
        const float* cur = prev + (size_t)N * F;
        float e[kMatchN][kMatchN];
#pragma unroll
        for (int a = 0; a < kMatchN; ++a)
#pragma unroll
          for (int j = 0; j < kMatchN; ++j) e[a][j] = (a < N && j < N) ? match_err(prev + slot[a < N ? a : 0] * F, cur + j * F, F) : 3.0e38f;
        int assign[kMatchN];
#pragma unroll
        for (int a = 0; a < kMatchN; ++a) assign[a] = 0;
        greedy_rounds(e, N, assign);
#pragma unroll
        for (int a = 0; a < kMatchN; ++a)
          if (a < N) slot[a] = assign[a];
      }
      __syncthreads();
      if (lane < N) r = slot[lane];
      __syncthreads();
    }
    if (lane < N) ibuf[t * N + lane] = r;
  }
  __syncthreads();
  for (int i = lane; i < T * N; i += 64) idx_out[(size_t)b * T * N + i] = ibuf[i];
}

// ---- three objects, '3_only': the walk through time as a composition of per-frame transition tables -------------------
// The slot contents before frame t are the objects of frame t-1 under one of the 6 permutations P of (0, 1, 2), and the
// distances the matcher looks at are D_t[P(a)][j] with D_t[i][j] = |x_{t-1,i} - x_{t,j}|^2 on the RAW object order -- a
// table every frame can form on its own.  So a lane takes a chunk of consecutive frames, runs the reference's rule
// (nearest object per slot; on a collision the slots pick greedily in slot order with column knock-out, stove.py:273-301)
// from each of the 6 possible incoming permutations, and gets its chunk's transition map (6 entries); the maps are
// composed across the lanes by a log-step scan, every lane then knows the permutation it starts from and replays its
// chunk once to emit idx.  Same float operations on the same values as the serial walk (bit-identical indices), but the
// 99 dependent frame steps of lane 0 (65 us at T = 100) become ~2 steps per lane.
struct Perm3 {
  int p0, p1, p2;
};
__device__ __forceinline__ Perm3 perm3_of(int k) {
  Perm3 r;
  r.p0 = k >> 1;
  const int lo = r.p0 == 0 ? 1 : 0, hi = r.p0 == 2 ? 1 : 2;
  r.p1 = (k & 1) ? hi : lo;
  r.p2 = (k & 1) ? lo : hi;
  return r;
}
__device__ __forceinline__ int perm3_id(const Perm3& p) { return 2 * p.p0 + (p.p1 > p.p2 ? 1 : 0); }
__device__ __forceinline__ float sel3(int i, float a, float b, float c) { return i == 0 ? a : (i == 1 ? b : c); }
// one frame: slots hold raw objects (p0, p1, p2) of the previous frame -> raw objects of this frame
__device__ __forceinline__ Perm3 match3_step(const float (&D)[3][3], const Perm3& p) {
  float e[3][3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    e[0][j] = sel3(p.p0, D[0][j], D[1][j], D[2][j]);
    e[1][j] = sel3(p.p1, D[0][j], D[1][j], D[2][j]);
    e[2][j] = sel3(p.p2, D[0][j], D[1][j], D[2][j]);
  }
  int idx[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    int best = 0;
    if (e[a][1] < e[a][best]) best = 1;
    if (e[a][2] < sel3(best, e[a][0], e[a][1], e[a][2])) best = 2;
    idx[a] = best;
  }
  if (idx[0] == idx[1] || idx[0] == idx[2] || idx[1] == idx[2]) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      int best = 0;
      if (e[a][1] < e[a][0]) best = 1;
      if (e[a][2] < sel3(best, e[a][0], e[a][1], e[a][2])) best = 2;
      idx[a] = best;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        e[r][0] = best == 0 ? 1e12f : e[r][0];
        e[r][1] = best == 1 ? 1e12f : e[r][1];
        e[r][2] = best == 2 ? 1e12f : e[r][2];
      }
    }
  }
  return Perm3{idx[0], idx[1], idx[2]};
}

template <int F>
__global__ __launch_bounds__(64) void match3_table_k(const float* __restrict__ feat, long long* __restrict__ idx_out, int B, int T) {
  extern __shared__ float mlds[];                 // [T*3*F] features, then [T*3] int indices
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n_feat = T * 3 * F;
  int* ibuf = reinterpret_cast<int*>(mlds + n_feat);
  for (int i = lane; i < n_feat; i += 64) mlds[i] = feat[(size_t)b * n_feat + i];
  __syncthreads();
  const int chunk = (T - 1 + 63) / 64;             // transitions t = 1 + lane * chunk .. (T - 1 of them in all)
  const int t0 = 1 + lane * chunk, t1 = t0 + chunk < T ? t0 + chunk : T;
  auto table = [&](int t, float (&D)[3][3]) {
    float pv[3][F], cv[3][F];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int f = 0; f < F; ++f) {
        pv[i][f] = (mlds[((t - 1) * 3 + i) * F + f] + 1.0f) * 0.5f;
        cv[i][f] = (mlds[(t * 3 + i) * F + f] + 1.0f) * 0.5f;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        float sacc = 0.0f;
#pragma unroll
        for (int f = 0; f < F; ++f) {
          const float d = pv[i][f] - cv[j][f];
          sacc += d * d;
        }
        D[i][j] = sacc;
      }
  };
  // the chunk's map: entry k (3 bits) = permutation id after the chunk when entering with permutation k
  int cur[6] = {0, 1, 2, 3, 4, 5};
  for (int t = t0; t < t1; ++t) {
    float D[3][3];
    table(t, D);
#pragma unroll
    for (int k = 0; k < 6; ++k) cur[k] = perm3_id(match3_step(D, perm3_of(cur[k])));
  }
  unsigned m = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) m |= (unsigned)cur[k] << (3 * k);
  // inclusive scan of the composition over the lanes: S_l = F_l o ... o F_0
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const unsigned o = (unsigned)__builtin_amdgcn_ds_bpermute(((lane - d) & 63) * 4, (int)m);
    if (lane >= d) {
      unsigned n = 0;
#pragma unroll
      for (int k = 0; k < 6; ++k) n |= ((m >> (3 * ((o >> (3 * k)) & 7u))) & 7u) << (3 * k);      // first the lower lanes' map, then mine
      m = n;
    }
  }
  const unsigned before = (unsigned)__builtin_amdgcn_ds_bpermute(((lane - 1) & 63) * 4, (int)m);
  Perm3 p = perm3_of(lane == 0 ? 0 : (int)(before & 7u));        // entering with the identity at t = 0
  if (lane == 0) {
    ibuf[0] = 0;
    ibuf[1] = 1;
    ibuf[2] = 2;
  }
  for (int t = t0; t < t1; ++t) {
    float D[3][3];
    table(t, D);
    p = match3_step(D, p);
    ibuf[t * 3 + 0] = p.p0;
    ibuf[t * 3 + 1] = p.p1;
    ibuf[t * 3 + 2] = p.p2;
  }
  __syncthreads();
  for (int i = lane; i < T * 3; i += 64) idx_out[(size_t)b * T * 3 + i] = ibuf[i];
}

}  // namespace stove
