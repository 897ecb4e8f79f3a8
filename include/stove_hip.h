/* libstove_hip.so -- C ABI of the MI355X (gfx950) STOVE hot path.
 *
 * The reference (jlko/STOVE) has no FFI: the path sits behind Python modules whose work is a
 * stream of ATen ops.  Each entry point below replaces one such op chain; the comment names
 * the reference lines (relative to the reference repo root) it stands in for.  The Python
 * modules in stove_amd/ (same names/signatures as the reference's) bind these with ctypes.
 *
 * Conventions:
 *   - plain pointers + sizes only; every pointer is DEVICE memory owned by the caller
 *     (PyTorch allocates), float32 unless noted; `stream` is a hipStream_t passed as void*.
 *   - functions only enqueue work on `stream`: no allocation, no free, no synchronisation;
 *     re-entrant and thread-safe; safe under hipGraph capture.  The library reads nothing from
 *     the environment; its only process state is listed in one place ("process state" below):
 *     the per-device fork stream, two measurement switches, the open event list, the profiler.
 *   - workspaces are sized by the matching *_ws_bytes() query and need no initialisation.
 *   - return value: 0 on success, otherwise the hipError_t of the failed launch
 *     (stove_error_string() names it).
 *   - all gradients are OVERWRITTEN, never accumulated.
 */
#ifndef STOVE_HIP_H
#define STOVE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int stove_abi_version(void);
const char* stove_error_string(int code);

/* Baked parameter tables of the two random SPNs (built by stove_amd/spn/rat_spn.py from the
 * reference-layout parameters: vector_list.L.i.{means,sigma_params,params}).
 * Object SPN (probabilistic_models.py:8-22; R=6 replicas, 4 leaves x S=25 pixels, G=K=10):
 *   obj_scope     int32 [R*4][S]        pixel index of leaf row i          (region_graph.py:54-95)
 *   obj_leaf_slot int32 [R][100]        L*S+i of pixel p inside replica r
 *   obj_coef      f32   [R*4][S][G][3]  (a,b,c): leaf log-density = sum_p w_p (a x^2 + b x + c)   (rat_torch.py:83-109)
 *   obj_wsum      f32   [R*2][G*G][K]   softmax(params, 0) of the 12 inner sum nodes            (rat_torch.py:202-222)
 *   obj_wroot     f32   [R][K*K]        softmax over all R*K*K root weights
 * Background SPN (probabilistic_models.py:25-39; R=3, two 512-pixel leaves, G=6):
 *   bg_side       int32 [R][1024]       0/1: which leaf of replica r owns pixel p
 *   bg_coef       f32   [R][1024][G][3]
 *   bg_wroot      f32   [R][G*G]        softmax over all R*G*G root weights, row j2*G+j1
 */
typedef struct StoveSpnTables {
  const int32_t* obj_scope;
  const int32_t* obj_leaf_slot;
  const float* obj_coef;
  const float* obj_wsum;
  const float* obj_wroot;
  const int32_t* bg_side;
  const float* bg_coef;
  const float* bg_wroot;
  const float* bg_dense;   /* optional: bg_coef as the operand image of the scene forward's leaf GEMM (stove_bg_dense), made
                            * ahead of time by the caller; NULL = stove_scene_fwd builds it itself at the head of its chain */
} StoveSpnTables;

/* bg_coef / bg_side -> the dense (3072 x 48, fragment-ordered) coefficient image StoveSpnTables.bg_dense points at;
 * it depends on the parameters only.  dense: stove_bg_dense_floats() floats. */
size_t stove_bg_dense_floats(void);
int stove_bg_dense(const int32_t* bg_side, const float* bg_coef, float* dense, void* stream);

/* Gradients w.r.t. the baked tables (same shapes as above). */
typedef struct StoveSpnTableGrads {
  float* obj_coef;
  float* obj_wsum;
  float* obj_wroot;
  float* bg_coef;
  float* bg_wroot;
} StoveSpnTableGrads;

/* ---- RatSpn.forward(inputs, marginalized) for the object SPN (rat_torch.py:354-357 as called
 * at supair.py:76).  inputs/marg: (n,100) row-major, marg may be NULL.  out: (n,).
 * `xw` is the saved activation for the backward: ceil(n/64)*100*2*64 floats. */
size_t stove_objspn_tile_floats(int n);
int stove_objspn_fwd(const StoveSpnTables* t, const float* inputs, const float* marg, float* xw, float* out,
                     int n, void* stream);
size_t stove_objspn_bwd_ws_bytes(int n);
/* d_inputs/d_marg (n,100) may be NULL. */
int stove_objspn_bwd(const StoveSpnTables* t, const float* marg, const float* xw, const float* out, const float* dout,
                     float* d_inputs, float* d_marg, StoveSpnTableGrads* g, void* ws, int n, void* stream);

/* ---- RatSpn.forward for the background SPN (supair.py:67).  inputs/marg: (n,1024).
 * `ell` is the saved activation: stove_bgspn_saved_floats(n) floats. */
size_t stove_bgspn_saved_floats(int n);
int stove_bgspn_fwd(const StoveSpnTables* t, const float* inputs, const float* marg, float* ell, float* out,
                    int n, void* stream);
size_t stove_bgspn_bwd_ws_bytes(int n);
int stove_bgspn_bwd(const StoveSpnTables* t, const float* inputs, const float* marg, const float* ell, const float* out,
                    const float* dout, float* d_inputs, float* d_marg, StoveSpnTableGrads* g, void* ws, int n,
                    void* stream);

/* The same operator over n_pix input dimensions (the reference builds the background SPN over c x w x h, probabilistic_models.py:25-39;
 * its stock gravity / multibilliards frames are 50 x 50): tables bg_side [R][n_pix], bg_coef [R][n_pix][G][3], bg_wroot [R][G*G];
 * inputs / marg / d_inputs / d_marg (n, n_pix).  General-size kernels (csrc/spn_bg_generic.hip), not the tuned 1024-pixel ones. */
size_t stove_bgspn_saved_floats_d(int n, int n_pix);
int stove_bgspn_fwd_d(const StoveSpnTables* t, const float* inputs, const float* marg, float* ell, float* out, int n, int n_pix, void* stream);
size_t stove_bgspn_bwd_ws_bytes_d(int n, int n_pix);
int stove_bgspn_bwd_d(const StoveSpnTables* t, const float* inputs, const float* marg, const float* ell, const float* out, const float* dout,
                      float* d_inputs, float* d_marg, StoveSpnTableGrads* g, void* ws, int n, int n_pix, void* stream);

/* ---- Supair.likelihood (supair.py:44-110), fused: masks_from_z (:278-356), patches_from_z
 * (:241-276), both SPN sweeps, patch scaling + Exponential(beta) overlap prior (:79-94).
 * frames: (n_frames,1024) one-channel frames; z: (n_frames*n_obj,4) = [sx,sy,x,y].
 * seq_frames / seq_stride: the frames may be a time-slice of longer clips handed over WITHOUT a copy (Stove.forward scores
 * x[:, 1:], stove.py:731-736): frame f is row (f / seq_frames) * seq_stride + f % seq_frames behind `frames`; 0, 0 = dense.
 * ll: (n_frames,) log p(x,z); parts: (n_frames,3) = bg, patches, overlap (may be NULL).
 * saved: stove_scene_saved_floats() floats kept for the backward.  The object SPN's backward is linear in the upstream
 * gradient of a glimpse, so the forward runs it at once, at a gradient of 1, while the sample's leaf and sum-node values are
 * still in registers, and leaves the per-glimpse scratch (2 904 B) in `saved`; stove_scene_bwd* applies dL/d(glimpse
 * likelihood) where that scratch is consumed.  A caller that will never differentiate uses stove_scene_fwd_from(.., with_grad 0)
 * with the smaller buffer of stove_scene_fwd_floats(.., 0): same ll / parts bit for bit, no backward from it. */
size_t stove_scene_saved_floats(int n_frames, int n_obj);
size_t stove_scene_fwd_floats(int n_frames, int n_obj, int with_grad);     /* with_grad != 0: == stove_scene_saved_floats */
int stove_scene_fwd(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                    int seq_stride, float overlap_beta, float* ll, float* parts, float* saved, void* stream);
size_t stove_scene_bwd_ws_bytes(int n_frames, int n_obj);
/* dll: (n_frames,) -> dz: (n_frames*n_obj,4) and table gradients. */
int stove_scene_bwd(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                    int seq_stride, float overlap_beta, const float* saved, const float* dll, float* dz, StoveSpnTableGrads* g,
                    void* ws, void* stream);
/* The same with the parameter-gradient passes (leaf / sum / root table gradients: they only feed the optimiser) on a
 * second stream, so that they overlap with what the caller enqueues on `stream` next -- in STOVE the latency-bound
 * backward of the recursion, which leaves most of every CU idle.  dz is complete in `stream` order when the call
 * returns; *g is complete in `param_stream` order.  param_stream NULL or == stream: identical to stove_scene_bwd. */
int stove_scene_bwd_overlap(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                    int seq_stride, float overlap_beta, const float* saved, const float* dll, float* dz, StoveSpnTableGrads* g,
                    void* ws, void* stream, void* param_stream);
/* ---- RatSpn.forward / backward of the OBJECT SPN for any glimpse size and vector widths (reference probabilistic_models.py:8-22
 * with config.patch_width / patch_height / obj_spn_num_gauss / obj_spn_num_sums other than 10 / 10 / 10 / 10, config.py:99-100,
 * 119-120; the entry points above are the tuned kernels of the default shape).  R replicas (children of the root sum), each the
 * product of two sum vectors of S nodes over the G x G product of two Gaussian leaves; D dimensions, leaves of up to Lmax pixels.
 *   lscope (R*4, Lmax) int32: pixels of leaf (r, 2 j + side), padded with -1;   slot (R, D) int32: l * Lmax + i of pixel p in replica r
 *   coef (R*4, Lmax, G, 3) = (a, b, c) per pixel and component;  wsum (R*2, G*G, S) softmaxed over G*G;  wroot (R, S*S) softmaxed
 * marg NULL = nothing marginalised.  saved: stove_objspn_saved_floats_any floats; ws: stove_objspn_bwd_ws_bytes_any bytes.
 * d_inputs / d_marg may be NULL; g_* are overwritten.  Limits: R <= 8, G, S <= 16, D <= 1024. */
size_t stove_objspn_saved_floats_any(int n, int R, int G, int S, int D, int Lmax);
size_t stove_objspn_bwd_ws_bytes_any(int n, int R, int G, int S, int D, int Lmax);
int stove_objspn_fwd_any(const float* inputs, const float* marg, const int* lscope, const float* coef, const float* wsum,
                         const float* wroot, float* saved, float* out, int n, int R, int G, int S, int D, int Lmax, void* stream);
int stove_objspn_bwd_any(const float* inputs, const float* marg, const int* lscope, const int* slot, const float* coef,
                         const float* wsum, const float* wroot, const float* saved, const float* dout, float* d_inputs,
                         float* d_marg, float* g_coef, float* g_wsum, float* g_wroot, void* ws, int n, int R, int G, int S, int D,
                         int Lmax, void* stream);
/* ---- the reference's fixed-Gaussian debug models (SimpleBG / SimpleObj, probabilistic_models.py:42-90; config.debug_bg_model /
 * debug_obj_spn): out[r] = sum_p (1 - marg[r][p]) log Normal(mean, scale)(x[r][p]), marg unclamped as there; dx / dm may be NULL. */
int stove_gauss_ll_fwd(const float* x, const float* marg, float* out, int n, int d, float mean, float scale, void* stream);
int stove_gauss_ll_bwd(const float* x, const float* marg, const float* dout, float* dx, float* dm, int n, int d, float mean, float scale,
                       void* stream);
/* The same likelihood for any frame size and either sampling convention of the spatial transformer (reference supair.py:44-110 with
 * config width/height != 32 -- the reference's stock gravity / multibilliards data are 50 x 50, envs.py:771-773 -- or the
 * torch-1.0.1 convention align_corners=True its published runs used).  frames: rows of W*H floats (W = the size of the last
 * tensor dimension, the x direction of grid_sample); 10 x 10 glimpses and single-channel frames as above.  The object side runs the
 * 32 x 32 path's kernels with the geometry at run time; the background side the general-size SPN operator (stove_bgspn_fwd_d) on
 * the closed-form mask.  t->bg_coef is [3][W*H][6][3], t->bg_side (W*H) int32, t->bg_dense unused.  saved / ws sizes below;
 * *g complete in `param_stream` order (NULL = `stream`), dz in `stream` order. */
size_t stove_scene_saved_floats_any(int n_frames, int n_obj, int n_pix, int with_grad);
size_t stove_scene_bwd_ws_bytes_any(int n_frames, int n_obj, int n_pix);
int stove_scene_fwd_any(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                        int seq_stride, int W, int H, int align_corners, float overlap_beta, float* ll, float* parts, float* saved,
                        void* stream, int with_grad);
int stove_scene_bwd_any(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                        int seq_stride, int W, int H, int align_corners, float overlap_beta, const float* saved, const float* dll,
                        float* dz, StoveSpnTableGrads* g, void* ws, void* stream, void* param_stream);
/* The same with the stream the internal background-SPN chain forks from given explicitly (NULL = `stream`).  For callers that run the
 * scene calls on a stream which is itself a fork inside a hipGraph capture: pass the capture's origin stream, where frames, z, saved
 * and dll must then be ready (the HIP 7.0 runtime cannot end a capture in which two forked streams wait on each other). */
int stove_scene_fwd_from(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                         int seq_stride, float overlap_beta, float* ll, float* parts, float* saved, void* stream, void* fork_from,
                         int with_grad);
int stove_scene_bwd_from(const StoveSpnTables* t, const float* frames, const float* z, int n_frames, int n_obj, int seq_frames,
                         int seq_stride, float overlap_beta, const float* saved, const float* dll, float* dz,
                         StoveSpnTableGrads* g, void* ws, void* stream, void* param_stream, void* fork_from);

/* Stove.object_embedding (stove.py:565-590): emb (n_frames*n_obj, channels) = mean over the 10x10 glimpse of object k of
 * colour frame x_color (n_frames, channels, 32, 32); z (n_frames*n_obj, 4) = [sx, sy, x, y].  No gradients. */
int stove_glimpse_mean(const float* x_color, const float* z, float* emb, int n_frames, int n_obj, int channels, void* stream);

/* ---- MPE reconstruction of glimpses by the object SPN (Supair.spn_mpe, supair.py:382-424, with RatSpn.reconstruct,
 * rat_torch.py:359-372): inputs (n,100) glimpses, no marginalisation.  Every sum node keeps its best child
 * (argmax_k child_k + log w_k, first on ties); the walk from the root selects one replica and one Gaussian per leaf;
 * out (n,100) = those components' means on their scopes, clamped to [0,1].  leaf_means: f32 [R*4][25][10], leaf order of
 * obj_scope.  pick (n,5) int32 = (replica, component of each of its 4 leaves), may be NULL.  `xw`: scratch of
 * stove_objspn_tile_floats(n) floats.  No gradients. */
int stove_objspn_mpe(const StoveSpnTables* t, const float* leaf_means, const float* inputs, float* xw, float* out,
                     int32_t* pick, int n, void* stream);

/* ---- frame rendering (Supair.reconstruct_from_z, supair.py:484-498): out (n_frames,1024) =
 * clamp(bg + sum_k grid_sample(patch_k, inverse transform of z_k), 0, 1).  bg (1024,), z (n_frames*n_obj,4) = [sx,sy,x,y],
 * patches (.,100): row (f / frames_per_patch) * n_obj + k; frames_per_patch == 0: the single row 0 for every object. */
int stove_render_frames(const float* bg, const float* patches, int frames_per_patch, const float* z, float* out, int n_frames,
                        int n_obj, void* stream);

/* ---- glimpses + masks alone (supair.py:241-356), for the Supair.patches_from_z /
 * masks_from_z API: patches, marg_patch: (n_frames*n_obj,100); overlap: (n_frames*n_obj,).
 * `tile` is scratch of stove_objspn_tile_floats(n_frames*n_obj) floats. */
int stove_scene_glimpses(const float* frames, const float* z, int n_frames, int n_obj, float* tile,
                         float* patches, float* keep, void* stream);

/* ---- Dynamics.forward / core (dynamics.py:181-265): one GNN step.
 * params: the parameter image of stove_gnn_param_floats() floats built by
 * stove_amd/video_prediction/dynamics.py + ops._gnn_image (W | W^T | vectors | W packed | W^T packed: the last two hold every layer
 * once more in the [K/4][OUT][4] order the small-graph recursion kernels copy into LDS; layout in csrc/gnn.hip).
 * s_in (B,N,sin_dim): [state 16 | action embedding 4 | appearance 3] as configured (16 <= sin_dim <= 32);
 * result (B,N,32) = means|stds (dynamics.py:216), pred (B,N,32) = dynamic_pred (:208), may be NULL.
 * elu: 0 = leaky_relu(0.01) (the reference default, dynamics.py:109), 1 = elu. */
size_t stove_gnn_param_floats(void);
size_t stove_gnn_grad_floats(void);
int stove_gnn_blocks(int B, int N);
int stove_gnn_fwd(const float* s_in, const float* params, float* result, float* pred, int B, int N, int sin_dim,
                  int lim_enc, int elu, void* stream);
size_t stove_gnn_bwd_ws_bytes(int B, int N);
/* g_params: gradient image of stove_gnn_grad_floats() floats (W layout + vectors). d_pred may be NULL. */
int stove_gnn_bwd(const float* s_in, const float* params, const float* d_result, const float* d_pred, float* d_s_in,
                  float* g_params, void* ws, int B, int N, int sin_dim, int lim_enc, int elu, void* stream);

/* debug/measurement: one forward+backward GNN step that also writes the cycle counter at every stage
 * boundary of workgroup 0 into stamps[0..63] (int64, device). */
int stove_gnn_debug_stamps(const float* s_in, const float* params, const float* d_result, float* d_s_in, void* ws,
                           long long* stamps, int B, int N, int sin_dim, int lim_enc, int elu, void* stream);

/* ---- the inference recursion of Stove.stove_forward (stove.py:696-713) with Dynamics.constrain_z_dyn
 * (dynamics.py:147-179) and Stove.full_state (stove.py:103-170) fused, all Ts = T-skip steps in one launch.
 * z1 (B,N,18) state at t=skip-1 [sx,sy/sx,x,y,vx,vy,latent]; zsup,zsstd (B,Ts,N,6) SuPAIR means/stds for
 * t=skip..T-1; eps (B,Ts,N,18) standard-normal draws; extra (B,Ts,N,sin_dim-16) or NULL.
 * outputs (B,Ts,N,.): z 18, zdyn 16, zdstd 16, mean 18, std 18, pred 32 (NULL to skip). */
/* act: saved activations (stove_dynloop_act_floats() floats, ~9 KB per sequence-step at N=3), written by the forward and
 * read by the backward.
 * DOMAIN (2 <= N <= 6, forward and rollout): layers 2 and 3 of the relation / attention chains multiply on IEEE-half hi / lo pieces
 * (three v_mfma_f32_16x16x32_f16 per product): their input activations must stay below 65 504 in magnitude (beyond it the pieces are
 * inf and the outputs inf / NaN -- never finite garbage); an operand keeps max(2^-22 |x|, 2^-25) of absolute precision, i.e. weights
 * below 1/4 lose relative, not absolute, accuracy (tests/test_gpu_dynamics.py::test_recursion_forward_chains_domain_*).
 * The backward normalises every gradient column by a power of two first and has no such bound.
 * With 2 <= N <= 6 objects the small-graph kernels run (csrc/gnn_small*.hip: one wave per node row up to
 * four objects, one HALF-wave per node row and two tiles of edge columns for five and six): act is then REQUIRED by the backward
 * (the layout is a set of per-sequence streams), and the backward consists of a T-serial data-gradient launch plus a
 * weight-gradient launch over the streams; its workspace is sized by stove_dynloop_bwd_ws_bytes_ts.  For N > 6 (MFMA kernels of
 * csrc/gnn.hip) act may be NULL in both calls: the backward then recomputes each step. */
size_t stove_dynloop_act_floats(int B, int Ts, int N);
int stove_dynloop_fwd(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                      const float* params, float* z, float* zdyn, float* zdstd, float* mean, float* std_, float* pred,
                      float* act, int B, int Ts, int N, int sin_dim, int lim_enc, int elu, float pos_var, float vel_std, float lat_std,
                      void* stream);
size_t stove_dynloop_bwd_ws_bytes(int B, int N);
size_t stove_dynloop_bwd_ws_bytes_ts(int B, int Ts, int N);   /* the one to use (covers the streamed per-step gradients) */
/* upstream gradients dz,dzdyn,dmean,dstd,dpred may each be NULL. */
int stove_dynloop_bwd(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                      const float* params, const float* z, const float* act, const float* dz, const float* dzdyn,
                      const float* dmean, const float* dstd, const float* dpred, float* dz1, float* dzsup, float* dzsstd, float* dextra,
                      float* g_params, void* ws, int B, int Ts, int N, int sin_dim, int lim_enc, int elu, float pos_var,
                      float vel_std, float lat_std, void* stream);

/* The same with the weight-gradient work (g_params) on a second stream; the data gradients (dz1, dzsup, dzsstd, dextra) are
 * complete in `stream` order, g_params in `param_stream` order (see stove_scene_bwd_overlap). */
int stove_dynloop_bwd_overlap(const float* z1, const float* zsup, const float* zsstd, const float* eps, const float* extra,
                              const float* params, const float* z, const float* act, const float* dz, const float* dzdyn,
                              const float* dmean, const float* dstd, const float* dpred, float* dz1, float* dzsup, float* dzsstd,
                              float* dextra, float* g_params, void* ws, int B, int Ts, int N, int sin_dim, int lim_enc, int elu,
                              float pos_var, float vel_std, float lat_std, void* stream, void* param_stream);


/* ---- reward head of the action-conditioned dynamics model (dynamics.py:62-70, 254-263):
 *   reward = sigmoid(head1(sum over objects of head0(dynamic_pred))),  head0 = Linear(32,32) ReLU Linear(32,32),
 *   head1 = Linear(32,16) ReLU Linear(16,8) ReLU Linear(8,1), one kernel each way (cl = 32).
 * pred (items, n_obj, 32): the dynamics core's `dynamic_pred` rows, items = sequences x steps; reward (items).
 * params: the ten tensors in module order, nn.Linear layout, one after the other (stove_reward_head_param_floats() = 2785 floats):
 *   head0.0.weight (32,32) | .bias | head0.2.weight (32,32) | .bias | head1.0.weight (16,32) | .bias | head1.2.weight (8,16) | .bias |
 *   head1.4.weight (1,8) | .bias.  saved: stove_reward_head_saved_floats(items, n_obj) floats written by the forward, read by the backward.
 * bwd: d_reward (items) -> d_pred (items, n_obj, 32) and g_params (2785 floats, same layout; fixed summation order);
 * ws: stove_reward_head_bwd_ws_floats(items) floats. */
size_t stove_reward_head_param_floats(void);
size_t stove_reward_head_saved_floats(int items, int n_obj);
size_t stove_reward_head_bwd_ws_floats(int items);
int stove_reward_head_fwd(const float* pred, const float* params, float* reward, float* saved, int items, int n_obj, void* stream);
int stove_reward_head_bwd(const float* pred, const float* params, const float* reward, const float* saved, const float* d_reward, float* d_pred,
                          float* g_params, float* ws, int items, int n_obj, void* stream);

/* y (rows, out) = x (rows, in) W^T + b for a narrow layer (in, out <= 64; the action embedding Linear(action_space, 4 N) of
 * dynamics.py:238-244), exact fp32 FMAs; b NULL = no bias; w_transposed != 0: W is (in, out) (the layer's backward dx = dy W). */
int stove_small_linear(const float* x, const float* W, const float* b, float* y, int rows, int in_dim, int out_dim, int w_transposed, void* stream);

/* ---- Stove.rollout (stove.py:777-861), mean prediction: z_last (B,N,18) [sx,sy,...] ->
 * z_pred (B,num,N,18); zstd (B,num,N,16) and pred (B,num,N,32) optional; extra (B,A,N,E) cycled (t % A). */
int stove_rollout_fwd(const float* z_last, const float* extra, const float* params, float* z_pred, float* zstd, float* pred,
                      int B, int num, int A, int N, int sin_dim, int lim_enc, int elu, float pos_var, float vel_std,
                      float lat_std, void* stream);

/* ---- Stove._3_only_match_objects / _greedy_match_objects / _volatile_match_objects
 * (stove.py:200-329, 432-514, 331-430): the T-serial nearest-neighbour re-ordering of objects.
 * feat (B,T,N,F) matching features in [-1,1] (positions [, appearance]); mode 0 = '3_only',
 * 1 = 'greedy', 2 = 'volatile', 3 = '3_only' forced through the frame-by-frame walk (the three-object '3_only' case
 * normally runs as a composition of per-frame transition tables, same indices).  idx (B,T,N) int64: current object assigned to slot a.
 * perm (B,T,N,N) f32, pre-zeroed, only written in mode 2 (may be NULL otherwise). */
int stove_match_objects(const float* feat, long long* idx, float* perm, int B, int T, int N, int F, int mode, void* stream);

/* ---- SuPAIR state pipeline and ELBO assembly (elementwise / short stencils in t; replaces ~180 ATen launches per step).
 * stove_supair_state_fwd: raw codes (n*T*o, 8) of the recognition LSTM ->
 *     zc (n,T,o,8) = constrain_zp (supair.py:112-149; span_low = 8 spans then 8 lows, HOST floats),
 *     idx (n,T,o) = object matching on the positions (mode as stove_match_objects; pos (n,T,o,2) is scratch),
 *     zfix (n,T,o,8) = gathered + fix_supair-smoothed [mean 4 | std 4] (stove.py:516-563; hits (n,T,o) u8 = its mask),
 *     zl / sl (n,T-skip,o,6) = z_sup_full / z_sup_std_full[:, skip:] with finite-difference velocities
 *     (stove.py:172-198), init6 (n,o,6) = z_sup_full[:, skip-1].
 *     codes == NULL: zc and idx are INPUTS (states already constrained, a matching already chosen) and only the gather /
 *     fix_supair / velocity stage runs -- what the reference's fix_supair fixtures exercise.
 * stove_supair_state_bwd: gradients of zfix / zl / sl / init6 (any may be NULL) -> g_codes; gfix_ws: n*T*o*8 floats.
 * stove_zall_fwd/bwd: z of the scene likelihood, frames 1..T-1: SuPAIR means before `skip`, sampled states after,
 *     [sx, sy/sx, x, y] -> [sx, sy, x, y] (stove.py:731-736); bwd writes every element of g_zfix and g_zs (n,T-skip,o,18).
 * stove_elbo_fwd: out3 = { mean_t,b(trans_lik + img_lik - log_q) + mean(img_lik_sup), mean trans_lik, mean log_q }
 *     (stove.py:738-748); lik (n,T-1); trans_std16 HOST floats; part_ws n*4 floats.  stove_elbo_bwd: g_out = device
 *     scalar dL/d out3[0]. */
int stove_supair_state_fwd(const float* codes, const float* span_low, float* zc, float* pos, long long* idx, float* zfix,
                           unsigned char* hits, float* zl, float* sl, float* init6, int n, int T, int o, int skip, int fix,
                           int mode, void* stream);
int stove_supair_state_bwd(const float* zc, const long long* idx, const unsigned char* hits, const float* zfix, const float* g_zfix,
                           const float* g_zl, const float* g_sl, const float* g_init6, const float* span_low, float* gfix_ws,
                           float* g_codes, int n, int T, int o, int skip, void* stream);
/* The same with the recursion's whole initial state as output: init (n, o, init_ld) gets the six SuPAIR values in columns 0..5 and,
 * with lat_noise (n, o, lat_dim) standard-normal draws, 0.01 x lat_noise in columns 6.. (stove.py:663-668); the backward reads
 * g_init with the same row stride. */
int stove_supair_state_fwd2(const float* codes, const float* span_low, float* zc, float* pos, long long* idx, float* zfix,
                            unsigned char* hits, float* zl, float* sl, float* init, int init_ld, const float* lat_noise, int lat_dim,
                            int n, int T, int o, int skip, int fix, int mode, void* stream);
int stove_supair_state_bwd2(const float* zc, const long long* idx, const unsigned char* hits, const float* zfix, const float* g_zfix,
                            const float* g_zl, const float* g_sl, const float* g_init6, int init_ld, const float* span_low, float* gfix_ws,
                            float* g_codes, int n, int T, int o, int skip, void* stream);
int stove_zall_fwd(const float* zfix, const float* zs, float* zall, int n, int T, int o, int skip, void* stream);
/* dz_in (NULL = none): the gradient zs receives from its other consumers (the ELBO's log q and transition terms; layout of zs): g_zs = the
 * likelihood's part + dz_in, so that the caller needs no separate accumulation pass. */
int stove_zall_bwd(const float* zfix, const float* zs, const float* g_zall, const float* dz_in, float* g_zfix, float* g_zs, int n, int T, int o,
                   int skip, void* stream);
int stove_elbo_fwd(const float* zs, const float* mean, const float* std_, const float* zdyn, const float* lik, const float* trans_std16,
                   float* part_ws, float* out3, int n, int T, int o, int skip, void* stream);
int stove_elbo_bwd(const float* zs, const float* mean, const float* std_, const float* zdyn, const float* trans_std16, const float* g_out,
                   float* g_zs, float* g_mean, float* g_std, float* g_zdyn, float* g_lik, int n, int T, int o, int skip, void* stream);

/* debug aid of the measurement tools: device buffer of 64 int64 cycle stamps ([4 waves][16 phases]) written by
 * workgroup 0 of the small-graph time loops in their last step; NULL (default) switches it off. */
void stove_debug_set_stamps(long long* device_buffer);

/* ---- flat parameter arena.  When all model parameters are views into ONE fp32 buffer (and their gradients
 * into one gradient buffer, which is also the data-parallel all-reduce bucket), the SPN tables and the GNN
 * parameter image are baked from / their gradients pushed into those buffers directly:
 *   stove_spn_bake      leaf (a,b,c) from means / sigma_params (rat_torch.py:85-99), softmax of the sum
 *                       weights (rat_torch.py:210), permuted into the kernel order
 *   stove_spn_bake_bwd  the chain rule of the same, ACCUMULATED into grad_arena
 *   stove_arena_gather / _scatter_add   image[i] = arena[src[i]] (src < 0: zero padding) and its adjoint.
 * Offsets are float offsets into the arena; obj_* follow the kernel's leaf / sum order. */
typedef struct StoveSpnArenaPlan {
  const int32_t* obj_mu;    /* [24] */
  const int32_t* obj_rho;   /* [24] */
  const int32_t* obj_sum;   /* [12] */
  const int32_t* bg_mu;     /* [6]  */
  const int32_t* bg_rho;    /* [6]  */
  const int32_t* bg_gidx;   /* [3*1024] leaf*512 + row of pixel p in replica r */
  int32_t obj_root, bg_root;
  float obj_vmin, obj_vmax, bg_vmin, bg_vmax;
} StoveSpnArenaPlan;
int stove_spn_bake(const float* arena, const StoveSpnArenaPlan* plan, float* obj_coef, float* obj_wsum, float* obj_wroot,
                   float* bg_coef, float* bg_wroot, void* stream);
int stove_spn_bake_bwd(const float* arena, const StoveSpnArenaPlan* plan, const StoveSpnTableGrads* g, float* grad_arena, void* stream);
int stove_arena_gather(const float* arena, const int32_t* src, float* image, int n, void* stream);
int stove_arena_scatter_add(const float* gimage, const int32_t* src, float* grad_arena, int n, void* stream);

/* out[j] = sum_c parts[c][j] in chunk order (n a multiple of 4): the split-K partials of the weight-gradient GEMMs. */
int stove_sum_chunks(const float* parts, float* out, size_t n, int chunks, void* stream);

/* ---- auxiliary streams.  Every entry point only enqueues on the stream(s) it is given -- with ONE exception: the scene calls
 * (stove_scene_fwd / _bwd and their variants) run their background-SPN chain next to the object-SPN chain on a per-device FORK stream
 * that the library creates on first use (hipStreamNonBlocking), forks from the call's stream and joins back before the call's last
 * kernel, so the caller sees single-stream semantics (stove_set_overlap(0): no fork at all).  A caller that wants to
 * own that stream -- several models per device, its own capture discipline -- hands it over here: stream != NULL = use this one,
 * NULL = run the chain on the call's stream; restore_default != 0 = back to the library-owned stream.  Thread-safe; takes effect for
 * calls enqueued afterwards.
 * ---- process state.  Everything the library keeps between calls: the fork streams above; the switch below (a measurement
 * aid; the Python binding sets it once from STOVE_NO_OVERLAP, the library itself never reads the environment); the open event list (stove_event_list_*); the profiler at the end of this header. */
int stove_set_fork_stream(int device, void* stream, int restore_default);
/* on = 0: no internal fork stream, every chain of a scene call on the call's stream (default 1). */
int stove_set_overlap(int on);
/* process state, test / measurement switch: which glimpse-tile kernel the scene forward runs -- 1 (default): one lane per glimpse PIXEL with
 * the tile transposed through LDS for up to three objects, one lane per glimpse (the round-5 kernel) beyond; 0: lane per glimpse always;
 * 2: lane per pixel always.  Same numbers bit for bit. */
int stove_set_tile_lds(int mode);
/* ---- stream ordering and graph replay (no reference counterpart: the reference enqueues ~6000 ATen launches per step from Python,
 * train.py:443-473; here a step is ~70 launches that a trainer replays as captured hipGraphs, stove_amd/graphed.py).
 * stove_stream_after: everything enqueued on `to` from now on runs after everything enqueued on `from` so far.  Plain streams: an
 * event record + wait.  Both streams capturing in the SAME capture: the usual fork / join edge.  Capturing in two DIFFERENT captures:
 * an event-record node in `from`'s graph and an event-wait node in `to`'s graph on a persistent event -- the graphs then order
 * against each other at replay, provided the recording graph is launched first.  The *_overlap entry points order their streams
 * with exactly this call, so they may be captured either way.
 * stove_capture_begin / _end: capture what is enqueued on `stream` (relaxed mode) into a graph (n_nodes optional);
 * stove_graph_instantiate: graph -> executable graph (consumes the graph; exec_out NULL-valued for an empty graph);
 * stove_graph_launch / _destroy: replay / free it.
 * stove_event_list_begin / _end / _destroy: who owns the events of the two-capture case.  begin opens a list (one at a time per
 * process) that every such event created from now on is appended to, end closes it (returns the number of events), destroy
 * frees the events -- call it when the graphs that hold their nodes are gone.  Without an open list the events are leaked. */
void* stove_event_list_begin(void);
int stove_event_list_end(void* list);
int stove_event_list_destroy(void* list);
int stove_stream_after(void* to, void* from);
int stove_capture_begin(void* stream);
int stove_capture_end(void* stream, void** graph_out, int* n_nodes);
int stove_graph_instantiate(void* graph, void** exec_out);
int stove_graph_launch(void* exec, void* stream);
int stove_graph_destroy(void* exec);

/* ---- reparameterisation noise (stove.py:667, 679, 146/167): out[0..n) <- standard normal draws, Philox-4x32-10 + Box-Muller, a pure
 * function of (state[0] = seed, state[1] = call number, element index); state: two 64-bit words in DEVICE memory, state[1] is advanced by
 * one behind the draw (a captured launch therefore replays with fresh noise; no host-side generator state).  out 16-byte aligned. */
int stove_noise_normal(float* out, size_t n, unsigned long long* state, void* stream);

/* bw_transform (reference utils.py): x (n_frames, channels, pixels) -> out (n_frames, pixels) = clamp(sum over channels, 0, 1). */
int stove_bw_transform(const float* x, float* out, int n_frames, int channels, int pixels, void* stream);
/* the same from an 8-bit frame store (load_data.py:60-113 keeps float frames on the host; here the training
 * set may live on the device as uint8 = round(255 v)): out = clamp(sum_c (x_c / 255), 0, 1), fp32. */
int stove_bw_transform_u8(const unsigned char* x, float* out, int n_frames, int channels, int pixels, void* stream);

/* out[c] = sum_r a[r][c] of a row-major (rows, cols) matrix, cols a multiple of 4 or <= 64 (bias gradients of the
 * recognition network: 25 600 x 1024, 76 800 x 50, 76 800 x 8); ws: stove_colsum_ws_floats(rows, cols) floats.  Fixed summation order. */
size_t stove_colsum_ws_floats(int rows, int cols);
int stove_colsum(const float* a, float* out, float* ws, int rows, int cols, void* stream);
/* out (M, N) = a^T b over the rows of a (rows, M), b (rows, N), M * N <= 256 (weight gradient of a narrow linear layer over
 * many rows); ws: stove_small_tn_ws_floats floats; fixed summation order. */
size_t stove_small_tn_ws_floats(int rows, int M, int N);
int stove_small_tn(const float* a, const float* b, float* out, float* ws, int rows, int M, int N, void* stream);
/* the same with a second output receiving the same sums (b_ih and b_hh of an LSTM) and accumulate != 0: added to the outputs */
int stove_colsum2(const float* a, float* out, float* out2, int accumulate, float* ws, int rows, int cols, void* stream);

/* One Adam / AMSGrad step of torch.optim.Adam (weight_decay 0; reference train.py:46-49, 431-473) over the flat arena,
 * with clip_grad_norm_(max_norm) folded in (clip != 0: gradients are scaled by min(1, max_norm / (||grads|| + 1e-6)) on the
 * fly; the norm is computed here, in a fixed order, and stored to grad_norm_out if that is not NULL).
 * torch's per-parameter semantics are kept: a SEGMENT is one parameter tensor (each starts on a float4 boundary);
 * seg_of4[i] = segment of floats 4i..4i+3 (numel/4 ints); seg_trainable[s] = requires_grad; seg_steps[s] = steps taken so
 * far (float, advanced here).  A segment steps only if it is trainable and its gradient slice has a non-zero element (what
 * `p.grad is None` means in a flat buffer) and uses ITS step count for the bias corrections; other segments are untouched.
 * ws: stove_flat_adam_ws_bytes(nseg) bytes, zeroed once by the caller before the first call.  max_exp_avg_sq NULL = plain
 * Adam.  hyper_dev != NULL: f32[5] = lr, beta1, beta2, eps, max_norm in device memory override the by-value arguments
 * (captured hipGraphs: kernel arguments are frozen at capture, the learning-rate schedule is not). */
/* clip: bit 0 = clip the gradient to max_norm; bit 1 = "strict zero gradients": a tensor that has once received a gradient keeps
 * stepping while its gradient is all zero (torch.optim.Adam on zero-FILLED .grad tensors, the zero_grad() of the torch 1.0.1 the
 * reference pins); default (bit clear): an all-zero gradient slice means `grad is None` (current torch's zero_grad(set_to_none=True)). */
size_t stove_flat_adam_ws_bytes(int nseg);
int stove_flat_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq, size_t numel,
                    const int* seg_of4, const unsigned char* seg_trainable, float* seg_steps, int nseg, void* ws, float* grad_norm_out,
                    const float* hyper_dev, float lr, float beta1, float beta2, float eps, float max_norm, int clip, void* stream);

/* ---- the dense products of RnnStates (encoder.py:43-57: nn.LSTM's x W_ih^T, h W_hh^T, fc1, and their gradients) as fp32
 * GEMMs on the bf16 matrix cores:  C[m][n] = sum_k a(m,k) b(n,k) (+ bias[n]) (+ add[m*ldc + n]),  a(m,k) = A[m*lda + k], or A[k*lda + m] when
 * a_kmajor (the operand is stored K-major, as dg / x are in the weight-gradient products dW = dg^T x); b likewise.
 * nsplit = 2: every fp32 operand element is split on the fly into bf16 hi + lo and the product is three bf16 MFMAs with fp32
 * accumulation (~2^-18 relative per product); nsplit = 3: the same with IEEE-half pieces (~2^-22: an fp32-grade product) for
 * operands inside half's range -- an element keeps max(2^-22 |x|, 2^-25) and must stay below 65504; B, the weights of the forward
 * products, is cut as 2^8 B (floor 2^-33, ceiling 256) and the epilogue shifts back -- K-contiguous, float4-addressable operands
 * only; nsplit = 1: plain bf16 operands (2^-8), fp32 accumulate.  A, B, bias, C stay fp32.
 * splitk > 1: K is cut into `splitk` slices computed by different workgroups into ws (stove_gemm_bf16_ws_floats floats), then
 * summed in slice order into C (needs ldc == N, bias and add NULL): fills the chip when M x N is small and K huge (weight gradients).
 * tile: workgroup tile, 0 = chosen by the library, 1 = 256 x 128 (8 waves), 2 = 128 x 128 (4 waves), 3 = 256 x 256 (8 waves of
 * 64 x 128; needs 136 KB of LDS: one workgroup per CU).
 * Operands / outputs whose leading dimension, contiguous extent or base address is not a multiple of 4 floats (fc1 of the
 * recognition network: 50 columns) are handled element-wise (slow path, meant for small operands); split-K needs an aligned C and
 * takes no add, except add == C (the product is accumulated into C, a gradient view) or, with 2..15 slices, a dense 16-byte aligned
 * M x N add term (added by the slice sum); a bias is allowed with 2..15 slices (added by
 * the slice sum). */
size_t stove_gemm_bf16_ws_floats(int M, int N, int splitk);
int stove_gemm_bf16(const float* A, const float* B, const float* bias, const float* add, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                    int a_kmajor, int b_kmajor, int nsplit, int splitk, int tile, float* ws, void* stream);

/* ---- gate math of RnnStates' LSTM (encoder.py:43-51, torch.nn.LSTM cell, gate order i,f,g,o); the GEMMs
 * around it are stove_gemm_bf16 (or library GEMMs).  gx (n,4H): gate pre-activations -- x W_ih^T + b_ih + b_hh, or already
 * the full sum with the recurrent term (the recurrent GEMM adds gx in its epilogue) and then gh NULL; gh (n,4H): h_prev W_hh^T
 * or NULL; c_prev (n,H) or NULL (zero state).  bwd: dh, dc_in (NULL = 0) -> dg (n,4H) (NULL: not stored), dc_out (n,H);
 * dgx_sum (n,4H) or NULL = dg of this step + the n_more gate gradients dg_more[m][n][4H] of other steps: the gradient of the
 * input projection shared by all steps, formed once (by the last backward step) and feeding dW_ih.
 * fast != 0: sigmoid / tanh on v_exp_f32 / v_rcp_f32 (absolute error ~1e-7) instead of an IEEE division and ocml's tanhf. */
int stove_lstm_cell_fwd(const float* gx, const float* gh, const float* c_prev, float* c, float* h, int n, int H, int fast, void* stream);
int stove_lstm_cell_bwd(const float* gx, const float* gh, const float* c_prev, const float* c, const float* dh,
                        const float* dc_in, float* dg, float* dc_out, float* dgx_sum, const float* dg_more, int n_more, int n, int H,
                        int fast, void* stream);
/* ---- the whole output head of RnnStates in one kernel each way (encoder.py:53-56), csrc/head_fused.hip: fc1, sigmoid and fc2 on the
 * matrix cores; H must be 256, HID <= 64, OUT 8.  fc1_split != 0 (the default path): fc1's forward products as three MFMAs on IEEE-half
 * hi / lo pieces (2^-22 of the value per product, as stove_gemm_bf16 nsplit = 3); 0: on the fp32 MFMA like everything else of the head.
 * fwd: h (rows, H) -> h1 = sigmoid(h W1^T + b1) (rows, HID) and codes (rows, OUT) = h1 W2^T + b2; W1 (HID, H), W2 (OUT, HID).
 * bwd: dcodes (rows, OUT), h1, h -> gh (rows, H) = dL/dh and the parameter gradients gW1 (HID, H), gb1 (HID), gW2 (OUT, HID),
 * gb2 (OUT); accumulate != 0: added to what the four tensors hold (gradient views of a flat arena).  ws:
 * stove_enc_head_bwd_ws_floats floats.  Fixed summation order.  frames > 0: h, h1, gh are step-major (row = step * frames + frame,
 * as the LSTM writes them) while codes / dcodes are frame-major (row = frame * steps + step, encoder.py:57); 0: same row order. */
int stove_enc_head_fwd(const float* h, const float* W1, const float* b1, const float* W2, const float* b2, float* h1, float* codes, int rows,
                       int H, int HID, int OUT, int frames, int fc1_split, void* stream);
size_t stove_enc_head_bwd_ws_floats(int rows, int HID);
int stove_enc_head_bwd(const float* dcodes, const float* h1, const float* h, const float* W1, const float* W2, float* gh, float* gW1,
                       float* gb1, float* gW2, float* gb2, int accumulate, float* ws, int rows, int H, int HID, int OUT, int frames, void* stream);

/* ---- measurement hooks (bench.py): when enabled, every kernel launch of this library is bracketed by
 * two HIP events recorded on the launch stream.  stove_profile_report() synchronises them, writes
 * "kernel\ttotal_ms\tcount\twall_ms\n" lines into buf (wall_ms: the time the kernel's launches cover -- less than their sum
 * where launches on two streams overlap) and clears the records; returns the bytes needed.
 * This is the only process-global state of the library and it is off by default. */
void stove_profile_enable(int on);
size_t stove_profile_report(char* buf, size_t cap);

/* test / debugging utility: n_words 32-bit words at p <- value (tests poison a captured step's memory between replays) */
int stove_fill_words(void* p, uint32_t value, size_t n_words, void* stream);

/* self-test hooks used by tests/ (wave reduction) */
int stove_selftest_wave_sum(const float* in, float* out, int n_waves, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STOVE_HIP_H */
