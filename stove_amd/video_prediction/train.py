"""Training loop (reference model/video_prediction/train.py: AbstractTrainer :19-179, Trainer :182-848).

Same public surface -- `Trainer(config, stove, train_dataset, test_dataset)` with `train`, `test`,
`long_rollout`, `prediction_error`, `error_and_log`, `save`, `load`, `load_encoder` -- and the same
optimisation recipe (Adam+amsgrad, exponential lr decay with a floor, global-norm clipping at 1,
BCE reward term ramped in for action-conditioned data).  Additions for the MI355X build:
one process per GPU with a single RCCL all-reduce of the flat gradient between backward and
clipping (stove_amd/parallel.py), and frames/s in the log.  Plotting / GIF rendering of the
reference (matplotlib, imageio) is visualisation and is not reproduced.
"""
import itertools
import os
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.optim as optim
from torch import nn
from torch.utils.data import DataLoader

from ..arena import ParamArena
from .load_data import DeviceClipLoader, ShardedDataLoader
from ..optim import FlatAdam
from ..parallel import broadcast_int, broadcast_tensors
from ..utils.utils import ExperimentLogger, bw_transform, settle_host_gc


class AbstractTrainer:
    def __init__(self, config, stove, train_dataset, test_dataset):
        self.stove = stove
        self.params = stove.parameters()
        if config.debug_test_mode:
            config.print_every = 1
            config.plot_every = 1
        self.c = config
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self._graphed = None
        self.dataloader = train_dataset
        self.test_dataset = test_dataset
        self.test_dataloader = self._make_loader(test_dataset, test=True)
        self.optimizer = optim.Adam(self.stove.parameters(), lr=self.c.learning_rate, amsgrad=self.c.debug_amsgrad)
        if self.c.load_encoder is not None:
            self.load_encoder()
        if not self.c.supair_grad:
            self.disable_supair_grad()
        # [amd] flat parameter / gradient arena (one-launch table baking, gradient buffer == all-reduce bucket)
        self.bucket = ParamArena(self.stove, self.world_size)
        p0 = next(self.stove.parameters())
        if p0.is_cuda and p0.dtype == torch.float32:
            # same update rule and state-dict layout as the Adam above, one launch over the flat buffers
            self.optimizer = FlatAdam(self.bucket, lr=self.c.learning_rate, amsgrad=self.c.debug_amsgrad,
                                      strict_zero_grad=bool(getattr(self.c, 'strict_adam_zero_grad', False)))
        self.epoch_start, self.step_start = 0, 0
        if self.c.checkpoint_path is not None:
            self.load()
        else:
            self.sync_replicas()

    def sync_replicas(self):
        """[amd] Data parallelism: every rank starts from rank 0's parameters, optimiser state and step counters (each
        process initialised its own weights from its own torch RNG), and draws its reparameterisation noise from its own
        stream.  Called after construction and after load(); a no-op for a single process."""
        if self.world_size <= 1:
            return
        self.bucket.sync(0)
        if isinstance(self.optimizer, FlatAdam):
            broadcast_tensors(list(self.optimizer._flat.values()) + [self.optimizer._seg_steps], 0)   # moments AND per-tensor step counts
            self.optimizer._steps = broadcast_int(self.optimizer._steps)
            self.optimizer._bind()
        else:
            broadcast_tensors([v for st in self.optimizer.state.values() for v in st.values() if torch.is_tensor(v)], 0)
        self.epoch_start, self.step_start = broadcast_int(self.epoch_start), broadcast_int(self.step_start)
        seed = int(getattr(self.c, 'dp_seed', 0)) + self.rank
        torch.manual_seed(seed)                 # CPU and every device generator: eps of rank r = stream (dp_seed + r)
        self._resume_loader_epoch()

    def _resume_loader_epoch(self):
        """The per-epoch clip permutations are seeded dp_seed + epoch: a resumed run continues the sequence at epoch_start."""
        for ld in (getattr(self, '_train_dataset', None),):
            if ld is not None and hasattr(ld, 'epoch'):
                ld.epoch = int(self.epoch_start)

    @property
    def dataloader(self):
        return self._train_dataset

    @dataloader.setter
    def dataloader(self, train_dataset):
        self._train_dataset = self._make_loader(train_dataset)

    def _make_loader(self, dataset, test=False):
        """Shuffled, drop-last batches of clips (reference train.py:39-44, 69-76).  [amd] The set lives on the GPU and batches are
        gathered there (no per-step host collate / PCIe copy) when it fits the budget; otherwise the reference's DataLoader."""
        dev = torch.device(self.c.device)
        budget = float(getattr(self.c, 'device_dataset_gb', 64.0)) * 2 ** 30
        seed = int(getattr(self.c, 'dp_seed', 0))
        fs = self._frame_store()
        if dev.type == 'cuda' and getattr(self.c, 'device_dataset', True) \
                and DeviceClipLoader.nbytes(dataset, self.c.dtype, fs) <= budget:
            if test:        # evaluation runs on rank 0 alone: ITS loader covers the whole test set (no shard), shuffled like the reference's (train.py:39-44)
                return DeviceClipLoader(dataset, self.c.batch_size, dev, self.c.dtype, shuffle=True, drop_last=True, frame_store=fs)
            return DeviceClipLoader(dataset, self.c.batch_size, dev, self.c.dtype, shuffle=True, drop_last=True,
                                    rank=self.rank, world=self.world_size, seed=seed, frame_store=fs)
        if test:
            return DataLoader(dataset, batch_size=self.c.batch_size, shuffle=True, num_workers=self.c.num_workers, drop_last=True)
        if self.world_size > 1:
            return ShardedDataLoader(dataset, self.c.batch_size, self.c.num_workers, self.rank, self.world_size, seed)
        return DataLoader(dataset, batch_size=self.c.batch_size, shuffle=True, num_workers=self.c.num_workers, drop_last=True)

    # ------------------------------------------------------------------ checkpoints
    def _state(self, epoch, step):
        return {'epoch': epoch, 'step': step, 'model_state_dict': self.stove.state_dict(),
                'optimizer_state_dict': self.optimizer.state_dict()}

    def save(self, epoch, step):
        if self.rank != 0:
            return
        path = os.path.join(self.logger.checkpoint_dir, 'ckpt')
        torch.save(self._state(epoch, step), path + '_{:05d}'.format(step))
        torch.save(self._state(epoch, step), path)
        print('Parameters saved to {}'.format(self.logger.exp_dir))

    def load(self):
        ckpt = torch.load(self.c.checkpoint_path, map_location=self.c.device)
        if 'model_state_dict' in ckpt:
            self.stove.load_state_dict(ckpt['model_state_dict'])
            self.optimizer.load_state_dict(ckpt['optimizer_state_dict'])
            self.epoch_start, self.step_start = ckpt['epoch'], ckpt['step']
        else:
            self.stove.load_state_dict(ckpt)
        print('Parameters loaded from {}.'.format(self.c.checkpoint_path))
        self._resume_loader_epoch()
        self.sync_replicas()

    def load_encoder(self):
        pretrained = torch.load(self.c.load_encoder, map_location=self.c.device)['model_state_dict']
        own = self.stove.state_dict()
        picked = {k: v for k, v in pretrained.items() if k in own and 'encoder' in k}
        own.update(picked)
        self.stove.load_state_dict(own)
        print('Loaded the following supair parameters from {}:'.format(self.c.load_encoder))
        print(picked.keys())

    def disable_supair_grad(self):
        for p in self.stove.sup.parameters():
            p.requires_grad = False

    def init_t(self, tensor):
        return tensor.type(self.c.dtype).to(device=self.c.device)

    def init_images(self, tensor):
        """[amd] Frames for the model: uint8 frames of the 8-bit device store pass through (the step's first kernel converts them)."""
        if tensor.dtype == torch.uint8:
            return tensor.to(device=self.c.device)
        return self.init_t(tensor)

    def _frame_store(self):
        """[amd] config.frame_store: 'auto' keeps the bw plane (fp32, bit-identical model input) when the model only ever sees
        bw frames -- single-channel SPNs and no appearance features -- and colour fp32 otherwise; 'u8' / 'f32' / 'bw32' force one."""
        fs = getattr(self.c, 'frame_store', 'auto')
        bw_only = bool(self.c.debug_bw) and self.c.channels == 1 and not (self.c.debug_core_appearance or self.c.debug_match_appearance) \
            and self.c.dtype == torch.float32
        if fs == 'auto':
            fs = 'bw32' if bw_only else 'f32'
        if fs == 'bw32' and not bw_only:
            raise ValueError("frame_store='bw32' needs a model that consumes bw frames only (debug_bw, channels=1, no appearance features, float32)")
        if fs == 'bw32':
            self.c.input_bw_plane = True          # Stove.forward: (n, T, 1, w, h) inputs are already bw_transform(x)
        return fs

    def adjust_learning_rate(self, optimizer, value, step):
        lr = max(self.c.learning_rate * np.exp(-step / value), self.c.min_learning_rate)
        for group in optimizer.param_groups:
            group['lr'] = lr


def _save_clip(path, frames, fps=24):
    """frames (T, H, W) uint8 -> path.gif (PIL) or path.npy."""
    try:
        from PIL import Image
    except ImportError:
        np.save(path + '.npy', frames)
        return
    imgs = [Image.fromarray(f, mode='L') for f in frames]
    imgs[0].save(path + '.gif', save_all=True, append_images=imgs[1:], duration=int(1000 / fps), loop=0)


class Trainer(AbstractTrainer):
    def __init__(self, config, stove, train_dataset, test_dataset):
        super().__init__(config, stove, train_dataset, test_dataset)
        # [amd] one run directory per job: rank 0 logs and saves, the other ranks get a logger without a directory
        self.logger = ExperimentLogger(self.c) if self.rank == 0 else None
        self.z_types = ['z', 'z_sup', 'z_dyn'] if not self.c.supair_only else ['z']
        if self.c.action_conditioned:
            self.reward_loss = nn.MSELoss() if self.c.debug_mse else nn.BCELoss()

    # ------------------------------------------------------------------ metrics
    def prediction_error(self, predicted, true, return_velocity=True, return_id_swaps=True,
                         return_full=False, return_matched=False, level='sequence'):
        """Position / velocity error of (n,T,o,4) predictions against the labels under the best
        object permutation: one permutation per sequence (chosen on the first <=4 frames), or per
        image for SuPAIR-only training.  Also the number of sequences without identity swaps."""
        if self.c.supair_only:
            return_velocity, level = False, 'image'
        perms = list(itertools.permutations(range(self.c.num_obj)))
        perm_t = torch.tensor(perms, device=predicted.device)                     # (P, o)
        pos_p, pos_t = predicted[..., :2], true[..., :2]
        t_fit = min(4, predicted.shape[1])

        def permuted(x, best):
            """x (n, T, o, d) with the objects of sequence i reordered by perms[best[i]] (one gather, no host loop)."""
            idx = perm_t[best][:, None, :, None].expand(-1, x.shape[1], -1, x.shape[3])
            return x.gather(2, idx)
        if level == 'sequence':
            errs = torch.stack([torch.sqrt(((pos_p[:, :t_fit, list(p)] - pos_t[:, :t_fit]) ** 2).sum(-1)).mean((1, 2))
                                for p in perms], 1)
            best = errs.argmin(1)
            pos_m = permuted(pos_p, best)
        elif level == 'image':
            pf, tf = pos_p.flatten(end_dim=1), pos_t.flatten(end_dim=1)
            errs = torch.stack([torch.sqrt(((pf[:, list(p)] - tf) ** 2).sum(-1)).mean(1) for p in perms], 1)
            best = errs.argmin(1)
            pos_m = permuted(pf[:, None], best)[:, 0].reshape(pos_p.shape)
        else:
            raise ValueError
        res = {}
        dist_err = torch.sqrt(((pos_m - pos_t) ** 2).sum(-1))
        if return_full:
            per_t = dist_err.mean(-1)
            res['error'], res['std_error'] = per_t.mean(0).cpu(), per_t.std(0).cpu()
        else:
            per_seq = dist_err.mean((1, 2))
            res['error'], res['std_error'] = per_seq.mean().cpu(), per_seq.std().cpu()
        if return_velocity:
            vel_p = predicted[..., 2:4]
            vel_m = permuted(vel_p, best)
            v_err = torch.sqrt(((true[..., 2:] - vel_m) ** 2).sum(-1)).mean(-1)
            if return_full:
                res['v_error'], res['std_v_error'] = v_err.mean(0).cpu(), v_err.std(0).cpu()
            else:
                res['v_error'], res['std_v_error'] = v_err.mean().cpu(), v_err.std().cpu()
            if return_matched:
                res['vel_matched'] = vel_m
        if return_matched:
            res['pos_matched'] = pos_m
        if return_id_swaps:
            pf, tf = pos_p.flatten(end_dim=1), pos_t.flatten(end_dim=1)
            per_img = torch.stack([torch.sqrt(((pf[:, list(p)] - tf) ** 2).sum(-1)) for p in perms], 1)
            order = per_img.mean(-1).argmin(1).reshape(true.shape[:2])
            stable = (order[:, 1:] == order[:, :-1]).all(1)
            res['swaps'] = stable.sum().float().cpu() / true.shape[0]
        return res

    def error_and_log(self, elbo, reward, min_ll, prop_dict, data, step_counter, now, add=''):
        skip = self.c.skip
        perf = {'step': step_counter, 'time': now, 'elbo': elbo, 'reward': reward, 'min_ll': min_ll}
        perf.update({k: v for k, v in prop_dict.items() if k[0] != 'z'})
        z_true = self.init_t(data['present_labels'][:, skip:])
        for z in self.z_types:
            if z in ('z', 'z_sup'):
                predicted = prop_dict[z][..., 2:]
                scales = prop_dict[z].flatten(end_dim=2)[:, :2].mean(0)
                perf['scale_x'], perf['scale_y'] = scales[0], scales[1]
            else:
                predicted = prop_dict[z]
                perf['scale_x'] = perf['scale_y'] = float('nan')
            perf.update(self.prediction_error(predicted, z_true))
            for i, std in enumerate(prop_dict[z + '_std']):
                perf['z_std_{}'.format(i)] = std
            perf['type'] = z + add
            if self.rank == 0:
                self.logger.performance(perf)

    def _reward_term(self, rewards, data, step_counter, key='present_rewards', skip=True):
        target = self.init_t(data[key][:, self.c.skip:] if skip else data[key])
        loss = self.reward_loss(rewards.flatten(), target.flatten())
        ramp = self.c.debug_reward_rampup
        weight = min(1, step_counter / ramp) if ramp is not False else 1
        return loss, self.c.debug_reward_factor * weight * loss

    # ------------------------------------------------------------------ training
    def _graph_ok(self, step_counter):
        """config.graph_step (default on): replay the captured step when nothing but the loss is needed from it -- every
        step that does not log.  Data-parallel runs replay two graphs around the all-reduce; action-conditioned runs pass the
        reward targets and the ramped reward weight through device memory (stove_amd/graphed.py)."""
        if not getattr(self.c, 'graph_step', True):
            return False
        if not isinstance(self.optimizer, FlatAdam) or torch.device(self.c.device).type != 'cuda':
            return False
        if step_counter % self.c.print_every == 0 or step_counter % self.c.plot_every == 0:
            return False
        if self._graphed is None:
            from ..graphed import GraphedTrainStep
            self._graphed = GraphedTrainStep(self.stove, self.bucket, self.optimizer, 1 if self.c.debug_gradient_clip else None,
                                             self.c.supair_only, world_size=self.world_size,
                                             reward_loss=self.reward_loss if self.c.action_conditioned else None)
        return True

    def _graph_step(self, data, step_counter):
        """One replayed step on a batch dict -> (elbo, min_ll, mse_rewards) as device scalars (values of THIS step until the next call)."""
        images = self.init_images(data['present_images'])
        if not self.c.action_conditioned:
            elbo = self._graphed(images)
            return elbo, -1.0 * elbo, torch.zeros(1)
        actions = self.init_t(data['present_actions'])
        target = self.init_t(data['present_rewards'][:, self.c.skip:])
        ramp = self.c.debug_reward_rampup
        weight = self.c.debug_reward_factor * (min(1, step_counter / ramp) if ramp is not False else 1)
        elbo = self._graphed(images, actions, target, reward_weight=weight)
        rl = self._graphed.reward_value
        return elbo, -1.0 * elbo + weight * rl, rl

    def train_step(self, data, step_counter):
        """One optimisation step on a batch dict (present_images [, present_actions, present_rewards])."""
        images = self.init_images(data['present_images'])
        actions = self.init_t(data['present_actions']) if self.c.action_conditioned else None
        self.bucket.zero()
        elbo, prop_dict, rewards = self.stove(images, step_counter, actions, self.c.supair_only)
        mse_rewards = torch.zeros(1)
        if self.c.action_conditioned:
            min_ll = -1.0 * elbo
            mse_rewards, term = self._reward_term(rewards, data, step_counter)
            min_ll = min_ll + term
            min_ll.backward()
        else:
            # min_ll = -ELBO (train.py:452): seed the backward with -1 instead of building / differentiating the negation
            if getattr(self, '_minus_one', None) is None or self._minus_one.device != elbo.device:
                self._minus_one = torch.tensor(-1.0, device=elbo.device, dtype=elbo.dtype)
            elbo.backward(self._minus_one)
            min_ll = -elbo.detach()
        self.bucket.all_reduce()                     # [amd] one RCCL all-reduce of the flat gradient
        if isinstance(self.optimizer, FlatAdam):
            self.optimizer.step(max_norm=1 if self.c.debug_gradient_clip else None)      # clipping folded into the step
        else:
            if self.c.debug_gradient_clip:
                torch.nn.utils.clip_grad_norm_(self.stove.parameters(), 1)
            self.optimizer.step()
        return elbo, prop_dict, rewards, min_ll, mse_rewards

    def train(self, num_epochs=None):
        print('Starting training for {}'.format(self.c.description))
        print('Only pretraining.' if self.c.supair_only else 'Full inference.')
        step_counter = self.step_start
        start = time.time()
        if not self.c.supair_only:
            self.test(step_counter, start)
        num_epochs = self.c.num_epochs if num_epochs is None else num_epochs
        epoch = self.epoch_start
        try:
            epoch, step_counter = self._train_epochs(num_epochs, step_counter, start)
        finally:
            # the loader gathered its batches straight into the captured step's input while this loop ran (below): outside of it a
            # batch it hands out must not alias that buffer (a second consumer, the mcts loop swapping loaders, a direct batch() call)
            if hasattr(self.dataloader, 'present_images_out'):
                self.dataloader.present_images_out = None
        if not self.c.debug_test_mode and not self.c.supair_only:
            self.long_rollout(step_counter=step_counter)
        if not self.c.nolog and self.rank == 0:
            self.save(epoch, step_counter)
            open(os.path.join(self.logger.exp_dir, 'success'), 'w').close()
        print('Finished Training!')

    def _train_epochs(self, num_epochs, step_counter, start):
        epoch = self.epoch_start
        for epoch in range(self.epoch_start, num_epochs):
            for data in self.dataloader:
                now = time.time() - start
                step_counter += 1
                if self.c.debug_anneal_lr:
                    self.adjust_learning_rate(self.optimizer, self.c.debug_anneal_lr, step_counter)
                if self._graph_ok(step_counter):
                    # [amd] the whole step replayed as captured hipGraph(s) (stove_amd/graphed.py): one launch call instead of ~100,
                    # the host runs many steps ahead of the device; steps that log (they read prop_dict) stay eager
                    elbo, min_ll, mse_rewards = self._graph_step(data, step_counter)
                    prop_dict, rewards = self.stove.prop_dict, None
                    out = self._graphed.static_images()
                    if out is not None and hasattr(self.dataloader, 'present_images_out'):
                        self.dataloader.present_images_out = out     # later batches are gathered straight into the step's input
                else:
                    elbo, prop_dict, rewards, min_ll, mse_rewards = self.train_step(data, step_counter)
                    elbo, min_ll = elbo.detach(), min_ll.detach()      # values only from here on: drop the autograd graph
                if step_counter == self.step_start + 3:
                    settle_host_gc()            # [amd] the loop is warm: full collections of the long-lived objects stay out of it
                if step_counter % self.c.print_every == 0:
                    if self.world_size > 1:
                        # [amd] the logged ELBO / loss are the means over the global batch: one 3-float all-reduce
                        tot = torch.stack([elbo.reshape(()).float(), min_ll.reshape(()).float(),
                                           mse_rewards.reshape(()).float().to(elbo.device)])
                        dist.all_reduce(tot)
                        elbo, min_ll, mse_rewards = (tot / self.world_size).unbind(0)
                    if self.rank == 0:          # error metrics of rank 0's shard, as one process would log its batch
                        self.error_and_log(elbo.item(), mse_rewards.item(), min_ll.item(), prop_dict, data, step_counter, now)
                if step_counter % self.c.save_every == 0:
                    self.save(epoch, step_counter)
                if step_counter % self.c.long_rollout_every == 0:
                    self.long_rollout(idx=[0, 1])
                if self.c.debug_test_mode and not self.c.supair_only:
                    self.save(0, 0)
                    self.test(step_counter, now)
                    break
            if not self.c.supair_only:
                self.test(step_counter, start)
            print('Epoch: ', epoch, ' finished.')
            if self.c.debug_test_mode:
                break
        return epoch, step_counter

    @torch.no_grad()
    def test(self, step_counter, start):
        """ELBO + reconstruction errors on test clips, then rollout errors of the generative model."""
        if self.rank != 0:
            return                  # [amd] evaluation has no collective in it: rank 0 evaluates and logs, the others go on
        self.stove.eval()
        for i, data in enumerate(self.test_dataloader):
            now = time.time() - start
            present = self.init_images(data['present_images'])
            actions = future_actions = future_rewards = None
            if self.c.action_conditioned:
                actions = self.init_t(data['present_actions'])
                future_actions = self.init_t(data['future_actions'])
                future_rewards = self.init_t(data['future_rewards'])
            elbo, prop_dict, rewards = self.stove(present, self.c.plot_every, actions, self.c.supair_only)
            min_ll = -1.0 * elbo
            mse_rewards = torch.zeros(1)
            if self.c.action_conditioned:
                mse_rewards, term = self._reward_term(rewards, data, step_counter)
                min_ll = min_ll + term
            self.error_and_log(elbo.item(), mse_rewards.item(), min_ll.item(), prop_dict, data, step_counter, now, add='_roll')
            appearances = prop_dict['obj_appearances'][:, -1] if self.c.debug_core_appearance else None
            z_pred, rewards_pred = self.stove.rollout(prop_dict['z'][:, -1], actions=future_actions, appearance=appearances)
            future_reward_loss = 0
            if self.c.action_conditioned:
                future_reward_loss = self.reward_loss(rewards_pred.flatten(), future_rewards.flatten())
            perf = {'step': step_counter, 'time': now, 'elbo': elbo, 'reward': future_reward_loss}
            perf.update(self.prediction_error(z_pred[..., 2:], self.init_t(data['future_labels'])))
            perf.update({k: v for k, v in prop_dict.items() if k[0] != 'z'})
            perf['type'] = 'rollout'
            if self.rank == 0:
                self.logger.performance(perf)
            if self.c.debug_test_mode or i > 7:
                break
        self.stove.train()

    @torch.no_grad()
    def long_rollout(self, idx=None, actions=None, step_counter=None, num=500):
        """Roll the dynamics out for `num` frames from the first visible frames of a few test
        sequences, log the position error over time and render real / rollout / reconstruction clips with
        `reconstruct_from_z` (reference train.py:684-849; GIFs through PIL when it is importable, uint8 .npy otherwise)."""
        if self.rank != 0:
            return None
        self.stove.eval()
        idx = list(idx) if idx is not None else [0, 1]
        ds = self.test_dataset
        nv = self.c.num_visible
        present = self.init_t(torch.from_numpy(np.stack([ds.total_img[i, :nv] for i in idx])))
        act = self.init_t(torch.from_numpy(np.stack([ds.total_actions[i, :nv] for i in idx]))) \
            if self.c.action_conditioned else None
        elbo, prop_dict, _ = self.stove(present, self.c.plot_every, act, False)
        app = prop_dict['obj_appearances'][:, -1] if self.c.debug_core_appearance else None
        fut = None
        if self.c.action_conditioned:
            fut = self.init_t(torch.from_numpy(np.stack([ds.total_actions[i, nv:] for i in idx])))
        z_pred, _ = self.stove.rollout(prop_dict['z'][:, -1], num=num, actions=fut, appearance=app)
        avail = min(num, ds.total_data.shape[1] - nv)
        out = {'z_pred': z_pred.cpu().numpy()}
        if avail > 0:
            true = self.init_t(torch.from_numpy(np.stack([ds.total_data[i, nv:nv + avail] for i in idx])))
            err = self.prediction_error(z_pred[:, :avail, :, 2:], true, return_full=True, return_id_swaps=False)
            out.update({k: v.numpy() for k, v in err.items()})
        z_recon = prop_dict['z']
        if self.c.channels == 1:
            real = bw_transform(present) if present.shape[2] != 1 else present
            clips = {'real': real[:, self.c.skip:],
                     'rollout': self.stove.reconstruct_from_z(torch.cat([z_recon, z_pred], 1)),
                     'recon': self.stove.reconstruct_from_z(z_recon)}
            out.update({'frames_' + k: (255 * v).clamp(0, 255).to(torch.uint8).cpu().numpy() for k, v in clips.items()})
        if self.rank == 0 and not self.c.nolog:
            tag = 'final' if step_counter is None else '{:06d}'.format(step_counter)
            np.save(os.path.join(self.logger.rollout_states_dir, 'rollout_states_{}.npy'.format(tag)), out['z_pred'])
            for k in [k for k in out if k.startswith('frames_')]:
                _save_clip(os.path.join(self.logger.rollout_gifs_dir, k[len('frames_'):]), out[k][0, :, 0])
        self.stove.train()
        return out
