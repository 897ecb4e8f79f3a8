"""Dataset of sliding-window clips over simulated sequences (reference model/video_prediction/load_data.py:33-174).

`data` dicts are what the simulators in stove_amd/envs produce (or the reference's pickles):
X (episodes, frames, res, res, 3) or already channel-first, y (episodes, frames, n_obj, 4) and, for
action-conditioned data, action / reward / done.  Labels are rescaled to the model's [-1, 1]
frame; they only feed error metrics, never the loss.
"""
import pickle

import numpy as np
from torch.utils.data import Dataset


def load(path):
    ext = path.split('.')[-1]
    if ext == 'pkl':
        with open(path, 'rb') as f:
            return pickle.load(f)
    if ext == 'mat':
        from scipy.io import loadmat
        return loadmat(path)
    raise ValueError('File format {} not recognized.'.format(ext))


class StoveDataset(Dataset):
    def __init__(self, config, test=False, data=None):
        self.c = config
        if data is None:
            data = load(self.c.testdata if test else self.c.traindata)
        self.rl = 'action' in data
        action_space = data['action_space'] if self.rl else None
        img = np.asarray(data['X'])[:self.c.num_episodes]
        if img.shape[-1] == 3:                                   # (.., res, res, 3) -> (.., 3, res, res)
            img = np.transpose(img, (0, 1, 4, 2, 3))
        self.total_img = img
        if (data['y'].shape[0] < self.c.num_episodes) and not test:
            print('WARNING: Data shape smaller than num_episodes specified.')
        self.total_data = np.array(data['y'][:self.c.num_episodes], dtype=np.float64)
        if self.rl:
            self.total_actions = data['action'][:self.c.num_episodes]
            self.total_rewards = data['reward'][:self.c.num_episodes] + 1      # to [0, 1] for the BCE loss
            self.total_dones = data['done'][:self.c.num_episodes] if 'done' in data else np.zeros_like(self.total_rewards)
        coord_lim = data.get('coord_lim', data.get('hw', 10))
        self.data_info = {
            'width': img.shape[-1], 'height': img.shape[-2], 'r': data.get('r', 1.3), 'coord_lim': coord_lim,
            'action_space': action_space, 'action_conditioned': self.rl,
            'num_obj': self.total_data.shape[2], 'num_frames': self.total_data.shape[1]}
        if self.c.debug_add_noise and not test:
            self.total_data += 0.02 / 2 * coord_lim * np.random.normal(size=self.total_data.shape)
        self.total_data *= 2 / coord_lim                         # positions and velocities to the [-1, 1] frame
        self.total_data[..., :2] -= 1
        n_eps, n_frames = self.total_img.shape[:2]
        clip_len = (self.c.num_visible + self.c.num_rollout) * self.c.frame_step
        starts = n_frames - clip_len + 1
        ep, fr = np.meshgrid(np.arange(n_eps), np.arange(starts))
        self.idxs = np.stack([ep, fr], 2).reshape(-1, 2)

    def __len__(self):
        return len(self.idxs)

    def __getitem__(self, idx):
        step = self.c.frame_step
        i, j = self.idxs[idx]
        mid = j + self.c.num_visible * step
        end = mid + self.c.num_rollout * step
        sample = {'present_images': self.total_img[i, j:mid:step], 'future_images': self.total_img[i, mid:end:step],
                  'present_labels': self.total_data[i, j:mid:step], 'future_labels': self.total_data[i, mid:end:step]}
        if self.rl:
            for name, arr in (('actions', self.total_actions), ('rewards', self.total_rewards), ('dones', self.total_dones)):
                sample['present_' + name] = arr[i, j:mid:step]
                sample['future_' + name] = arr[i, mid:end:step]
        return sample


class DeviceClipLoader:
    """Training batches cut ON THE GPU from a device-resident copy of the dataset.

    The reference collates every batch on the host and copies it to the device (train.py:445-449): at the headline batch
    (256 clips x 100 colour frames) that is 314 MB per step, ~5.7 ms over PCIe -- as long as the whole GPU step.  A STOVE
    training set (1000 x 100 frames: 1.2 GB as fp32) is a rounding error in 288 GB of HBM, so it is uploaded once and a
    batch is ONE gather over a flat (episode * frame) index: no host work, no PCIe traffic per step.

    Yields the same dicts (keys, shapes) as `DataLoader(StoveDataset, batch_size, shuffle, drop_last=True)`, with tensors
    already on the device in the model dtype; the shuffle uses torch's global CPU generator like RandomSampler does.

    Data parallelism (`world` > 1): every rank holds the whole set (it is small) and draws the SAME
    permutation per epoch from `seed + epoch`; rank r owns `order[r::world]`, cut to a number of `batch_size` batches
    that is equal on all ranks.  `batch_size` is the PER-RANK batch, so the effective batch of a step is
    world x batch_size (weak scaling, what bench.py measures); `last_clip_ids` is what the last batch was cut from.
    """

    def __init__(self, dataset, batch_size, device, dtype, shuffle=True, drop_last=True, rank=0, world=1, seed=0, frame_store='f32'):
        """frame_store -- how the frames live in HBM (a compact device-resident frame store):
          'f32'   colour frames in the model dtype, what the reference's loader hands over (12 KB per 32x32 frame);
          'bw32'  the bw plane itself: bw_transform (utils.py:10-15), a deterministic per-frame function, applied ONCE at upload
                  and kept as fp32 -- bit-identical model input, 4 KB per frame, no per-step transform pass; batches carry
                  (n, T, 1, w, h) images and the model must be told (config.input_bw_plane, set by the Trainer);
          'u8'    colour frames as uint8 = round(255 v), 3 KB per frame; the first kernel of the step converts on load
                  (stove_bw_transform_u8).  Lossy for float renderings (|dv| <= 1/510 per channel), exact for 8-bit sources."""
        import torch
        if frame_store not in ('f32', 'bw32', 'u8'):
            raise ValueError("frame_store must be 'f32', 'bw32' or 'u8'")
        self.ds, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), shuffle, drop_last
        self.rank, self.world, self.seed, self.epoch = int(rank), int(world), int(seed), 0
        self.last_clip_ids = None
        self.frame_store = frame_store
        self.present_images_out = None        # optional (batch, nv, c, w, h) tensor that receives the gathered present_images
        c = dataset.c
        self.step, self.nv, self.nr = c.frame_step, c.num_visible, c.num_rollout

        def up(a):
            t = torch.as_tensor(np.ascontiguousarray(a)).to(device=device, dtype=dtype)
            return t.view(t.shape[0] * t.shape[1], *t.shape[2:])

        def up_images(img):
            if frame_store == 'f32':
                return up(img)
            if frame_store == 'u8':
                q = np.rint(np.clip(np.asarray(img, dtype=np.float64), 0.0, 1.0) * 255.0).astype(np.uint8)
                t = torch.as_tensor(np.ascontiguousarray(q)).to(device=device)
                return t.view(t.shape[0] * t.shape[1], *t.shape[2:])
            from ..utils.utils import bw_transform
            parts = []
            for e0 in range(0, img.shape[0], 64):           # 64 episodes at a time: the colour copy is transient
                x = torch.as_tensor(np.ascontiguousarray(img[e0:e0 + 64])).to(device=device, dtype=dtype)
                parts.append(bw_transform(x))
            t = torch.cat(parts, 0)
            return t.view(t.shape[0] * t.shape[1], *t.shape[2:])
        self.n_frames = dataset.total_img.shape[1]
        self.store = {'images': up_images(dataset.total_img), 'labels': up(dataset.total_data)}
        if dataset.rl:
            self.store.update(actions=up(dataset.total_actions), rewards=up(dataset.total_rewards), dones=up(dataset.total_dones))
        self.idxs = torch.as_tensor(dataset.idxs, dtype=torch.long)
        self.device = device
        self._present = torch.arange(self.nv, device=device) * self.step
        self._future = (self.nv + torch.arange(self.nr, device=device)) * self.step

    @staticmethod
    def nbytes(dataset, dtype, frame_store='f32'):
        import torch
        item = torch.empty((), dtype=dtype).element_size()
        img = dataset.total_img
        if frame_store == 'u8':
            n_img = img.size
        elif frame_store == 'bw32':
            n_img = img.size // img.shape[2] * item
        else:
            n_img = img.size * item
        n = dataset.total_data.size
        if dataset.rl:
            n += dataset.total_actions.size + dataset.total_rewards.size + dataset.total_dones.size
        return n_img + n * item

    def store_bytes(self):
        """Bytes of HBM the resident set occupies."""
        return sum(t.numel() * t.element_size() for t in self.store.values())

    def __len__(self):
        n = len(self.idxs) // self.world
        return n // self.batch_size if self.drop_last or self.world > 1 else (n + self.batch_size - 1) // self.batch_size

    def _base(self, clip_ids):
        """flat (episode * frame) index of the first frame of every clip, on the host"""
        ij = self.idxs[clip_ids]
        return ij[:, 0] * self.n_frames + ij[:, 1]

    def batch(self, clip_ids, base=None):
        """The batch of the given clip numbers (indices into dataset.idxs); `base`: their first-frame indices already on the device.
        While `present_images_out` is set (Trainer.train sets it to its captured step's input for the duration of its loop and
        clears it afterwards) every batch returns THAT tensor as present_images: such a batch is valid until the next one is made."""
        if base is None:
            base = self._base(clip_ids).to(self.device)
        pres, fut = base[:, None] + self._present, base[:, None] + self._future
        out = {}
        for name, t in self.store.items():
            dst = self.present_images_out if name == 'images' else None
            if dst is not None and dst.dtype == t.dtype and tuple(dst.shape) == (pres.shape[0], pres.shape[1]) + tuple(t.shape[1:]):
                import torch
                # straight into the caller's static buffer (the captured step's input): no second copy of the batch
                torch.index_select(t, 0, pres.reshape(-1), out=dst.view(-1, *t.shape[1:]))
                out['present_' + name] = dst
            else:
                out['present_' + name] = t[pres]
            out['future_' + name] = t[fut]
        return out

    def __iter__(self):
        import torch
        n = len(self.idxs)
        if self.world > 1:
            from ..parallel import shard_order
            g = torch.Generator().manual_seed(self.seed + self.epoch)         # the same order on every rank
            order = torch.randperm(n, generator=g) if self.shuffle else torch.arange(n)
            order = shard_order(order, self.rank, self.world, self.batch_size)
            self.epoch += 1
        else:
            order = torch.randperm(n) if self.shuffle else torch.arange(n)
        # ONE host-to-device copy per epoch (the clip order of the whole epoch), not one per batch: a pageable copy waits for the stream,
        # i.e. for the previous step -- it would tie the host to the device once per step and undo what the replayed step buys
        nb = len(self)
        base_all = self._base(order[:nb * self.batch_size] if self.drop_last or self.world > 1 else order).to(self.device)
        for b in range(nb):
            self.last_clip_ids = order[b * self.batch_size:(b + 1) * self.batch_size]
            yield self.batch(self.last_clip_ids, base_all[b * self.batch_size:(b + 1) * self.batch_size])


class ShardedDataLoader:
    """The host DataLoader fallback of the Trainer under data parallelism: a DistributedSampler with the shared seed
    (disjoint per-rank shards of one common permutation, equal batch counts), re-seeded every epoch."""

    def __init__(self, dataset, batch_size, num_workers, rank, world, seed):
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        self.sampler = DistributedSampler(dataset, num_replicas=world, rank=rank, shuffle=True, seed=seed, drop_last=True)
        self.loader = DataLoader(dataset, batch_size=batch_size, sampler=self.sampler, num_workers=num_workers, drop_last=True)
        self.epoch = 0

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        self.sampler.set_epoch(self.epoch)
        self.epoch += 1
        return iter(self.loader)
