"""Experiment configuration.

Same attribute names, defaults and override semantics as the reference's
model/video_prediction/config.py:6-134 (`StoveConfig` is a bag of class attributes that
`load_args` overrides from `--args key value ...` pairs or a restored config.txt), so configs
and run scripts interchange.  Attributes marked [amd] are additions of this build.
"""
import torch

_DEFAULTS = dict(
    # ---- experiment
    description='unnamed experiment', nolog=False, experiment_dir='./experiments/unsorted',
    checkpoint_path=None, keep_folder=False, action_conditioned=None, random_seed=None,
    supairvised=False, load_encoder=None, supair_only=False, supair_grad=True, debug_test_mode=False,
    # ---- data (the None entries are filled from the dataset by main.py)
    traindata='./data/billiards_train.pkl', testdata='./data/billiards_test.pkl',
    num_visible=8, num_rollout=8, frame_step=1, num_episodes=1000, num_frames=None,
    width=None, height=None, channels=1, num_obj=None, r=None, coord_lim=None, action_space=None,
    debug_add_noise=False,
    # ---- optimisation
    batch_size=256, cl=32, learning_rate=0.002, min_learning_rate=0.0002, debug_anneal_lr=40000.0,
    num_epochs=400, debug_amsgrad=True, debug_gradient_clip=True,
    # ---- runtime
    device=None, dtype=torch.double, max_threads=8, num_workers=4,
    # ---- logging
    debug=True, n_plot_sequences=5, print_every=100, plot_every=1e19, save_every=10000,
    long_rollout_every=10000, visdom=False, debug_extend_plots=False,
    # ---- STOVE
    skip=2, transition_lik_std=[0.01, 0.01, 0.01, 0.01], debug_fix_supair=True,
    debug_match_appearance=False, debug_no_latents=False,
    # ---- action-conditioned variant
    debug_reward_factor=15000, debug_reward_rampup=20000, debug_mse=False,
    debug_core_appearance=False, debug_appearance_dim=3,
    # ---- dynamics core
    debug_nonlinear='relu', debug_latent_q_std=0.04, debug_xavier=False,
    # ---- SPNs / SuPAIR
    debug_bw=True, patch_height=10, patch_width=10,
    obj_min_var=0.12, obj_max_var=0.35, bg_min_var=0.002, bg_max_var=0.16,
    scale_var=0.3, pos_var=0.3, min_obj_scale=0.1, max_obj_scale=0.8, min_y_scale=0.75, max_y_scale=1.25,
    obj_pos_bound=0.9, obj_spn_num_gauss=10, obj_spn_num_sums=10, overlap_beta=10.0,
    debug_bg_model=False, debug_obj_spn=False, debug_simple_bg_var=0.1, debug_simple_obj_var=0.2,
    debug_match_objects='3_only', debug_no_reuse=False, debug_no_velocity=False,
    # ---- [amd] additions
    world_size=1,            # data-parallel ranks (one process per GPU, RCCL all-reduce of the flat gradient)
    rank=0,                  # this process's rank
    dp_seed=0,               # base seed shared by all ranks: clip permutation of epoch e = dp_seed + e, noise = dp_seed + rank
    align_corners=False,     # spatial-transformer convention; False = what the runnable reference computes
    fused_dynamics=True,     # run the inference recursion in the persistent HIP time-loop kernel
    fused_state=True,        # constrain_zp / matching / fix_supair / velocities as the fused state pipeline (csrc/state.hip)
    fused_reward_head=True,  # action-conditioned model: the reward head as one HIP kernel each way (csrc/reward_head.hip) instead of ten library launches
    fused_elbo=True,         # log q, transition likelihood and the ELBO means in two launches
    graph_step=True,         # Trainer: replay the non-logging training steps as captured hipGraph(s) (stove_amd/graphed.py)
    frame_store='auto',      # DeviceClipLoader: 'auto' (bw plane as fp32 when the model only sees bw frames, else colour fp32), 'bw32', 'u8', 'f32'
    input_bw_plane=False,    # Stove.forward: single-channel input frames are already bw_transform(x) (set by the Trainer for frame_store bw32)
    device_dataset=True,     # Trainer: training set resident on the GPU, batches gathered there (load_data.DeviceClipLoader)
    device_dataset_gb=64.0,  # ... when it needs at most this much HBM
    device_noise='philox', strict_adam_zero_grad=False,   # [amd] reparameterisation draws on the GPU: 'philox' = the library's counter-based generator (device-resident state, drawn ahead on the parameter stream), 'torch' = torch.randn
    encoder_gemm='bf16x3',   # recognition-network GEMMs: 'bf16x3' (fp32 products as 3 sixteen-bit MFMAs on hi/lo pieces: IEEE-half pieces forward, bf16 pieces backward), 'fp32' (library), 'bf16'
)


class StoveConfig:
    """Completely specifies an experiment: data, training and model parameters."""


for _k, _v in _DEFAULTS.items():
    setattr(StoveConfig, _k, _v)
del _k, _v
