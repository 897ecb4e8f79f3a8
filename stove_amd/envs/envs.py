"""Numpy physics simulators that synthesise STOVE's training videos (CPU data synthesis).

Counterpart of the reference's model/envs/envs.py (PhysicsEnv :160-343, BillardsEnv :366-442,
GravityEnv :445-524, AvoidanceTask :108-113, MonteCarloActionPolicy :559-585) reduced to what
the hot path's benchmarks and tests need: ball rendering, billiards and gravity dynamics, the
avoidance task with its sticky random action policy.  Every simulator owns a private legacy
`np.random.RandomState(seed)`; for a given seed it consumes the stream exactly like the
reference consumes the global numpy stream after `np.random.seed(seed)`, so frames and states
are reproduced bit for bit (pinned by tests/golden/g0_envs.npz).
"""
import numpy as np

BALL_COLOURS = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [1, 0, 1], [0, 1, 1]])


def _row_norm(a):
    return np.linalg.norm(a) if a.ndim == 1 else np.linalg.norm(a, axis=1).reshape(-1, 1)


class PhysicsEnv:
    """n discs of radius r and mass m in a hw x hw box, rendered at res x res."""

    def __init__(self, n=3, r=1., m=1., hw=10, granularity=5, res=32, t=1., init_v_factor=None,
                 friction_coefficient=0., seed=None, use_colors=None, rng=None):
        self.rng = rng if rng is not None else np.random.RandomState(seed)
        self.n = n
        self.r = np.array([[r]] * n) if np.isscalar(r) else r
        self.m = np.array([[m]] * n) if np.isscalar(m) else m
        self.hw = hw
        self.internal_steps = granularity
        self.eps = 1 / granularity
        self.res = res
        self.t = t
        self.x = self.init_x()
        self.v = self.init_v(init_v_factor)
        self.a = np.zeros_like(self.v)
        self.fric_coeff = friction_coefficient
        self.use_colors = (n >= 3) if use_colors is None else use_colors

    # -- initial conditions ---------------------------------------------------------------
    def init_x(self):
        while True:
            x = self.rng.rand(self.n, 2) * self.hw / 2 + self.hw / 4
            inside = np.all(x - self.r >= 0) and np.all(x + self.r <= self.hw)
            apart = all(_row_norm(x[i] - x[j]) >= self.r[i] + self.r[j]
                        for i in range(self.n) for j in range(i))
            if inside and apart:
                return x

    def init_v(self, init_v_factor=None):
        v = self.rng.normal(size=(self.n, 2))
        v = v / np.sqrt((v ** 2).sum()) * .5
        if init_v_factor is not None:
            v = v * self.rng.uniform(1 / init_v_factor, init_v_factor)
        return v

    # -- dynamics -------------------------------------------------------------------------
    def simulate_physics(self, actions):
        raise NotImplementedError

    def step(self, action=None, mass_center_obs=False):
        acting = action is not None
        if acting:
            self.v[0] = action * self.t          # the controlled ball is driven directly
        for _ in range(self.internal_steps):
            self.x += self.t * self.eps * self.v
            if mass_center_obs:
                self.x += self.hw / 2 - np.sum(self.m * self.x, 0) / np.sum(self.m)
            self.v -= self.fric_coeff * self.m * self.v * self.t * self.eps
            self.v = self.simulate_physics(acting)
        return self.draw_image(), np.concatenate([self.x, self.v], axis=1), False

    # -- rendering ------------------------------------------------------------------------
    def get_obs_shape(self):
        return (self.res, self.res, 3)

    def draw_image(self):
        if self.n > 6:
            raise ValueError('Max self.n implemented currently is 6.')
        img = np.zeros((self.res, self.res, 3), dtype='float')
        centres = (0.5 / self.res + np.arange(0, 1, 1. / self.res, dtype='float')) * self.hw
        I, J = np.meshgrid(centres, centres)
        for i in range(self.n):
            blob = np.exp(-(((I - self.x[i, 0]) ** 2 + (J - self.x[i, 1]) ** 2) / (self.r[i] ** 2)) ** 4)
            if self.use_colors:
                for ch in range(3):
                    img[:, :, ch] += BALL_COLOURS[i, ch] * blob
            else:
                img[:, :, i % 3] += blob
        img[img > 1] = 1
        return img


class BillardsEnv(PhysicsEnv):
    """Elastic bouncing balls."""

    def __init__(self, n=3, r=1., m=1., hw=10, granularity=5, res=32, t=1., init_v_factor=None,
                 friction_coefficient=0., seed=None, use_colors=None, drift=False, rng=None):
        super().__init__(n, r, m, hw, granularity, res, t, init_v_factor, friction_coefficient, seed, use_colors, rng)
        self.collisions = 0
        self.drift = drift

    def simulate_physics(self, actions):
        v = self.v.copy()
        dt = self.eps * self.t
        for i in range(self.n):                      # walls
            for ax in range(2):
                nxt = self.x[i, ax] + v[i, ax] * dt
                ri = float(np.ravel(self.r[i])[0])
                if nxt < ri:
                    self.x[i, ax] = ri
                    v[i, ax] = -v[i, ax]
                elif nxt > self.hw - ri:
                    self.x[i, ax] = self.hw - ri
                    v[i, ax] = -v[i, ax]
        if self.drift:
            return v
        for i in range(self.n):                      # pairwise elastic collisions
            for j in range(i):
                gap = _row_norm((self.x[i] + v[i] * self.t * self.eps) - (self.x[j] + v[j] * self.t * self.eps))
                if gap < self.r[i] + self.r[j]:
                    controlled = actions and j == 0
                    if controlled:
                        self.collisions = 1
                    w = self.x[i] - self.x[j]
                    w = w / _row_norm(w)
                    v_i, v_j = np.dot(w.transpose(), v[i]), np.dot(w.transpose(), v[j])
                    if controlled:
                        v_j = 0
                    m1, m2 = self.m[i], self.m[j]
                    new_v_j = (2 * m1 * v_i + v_j * (m2 - m1)) / (m1 + m2)
                    new_v_i = new_v_j + (v_j - v_i)
                    v[i] += w * (new_v_i - v_i)
                    v[j] += w * (new_v_j - v_j)
                    if controlled:
                        v[j] = 0
        return v

    def step(self, action=None):
        self.collisions = 0
        return super().step(action)


class GravityEnv(PhysicsEnv):
    """Mutually attracting balls, simulated in the centre-of-mass frame."""

    def __init__(self, n=3, r=1., m=1., hw=10, granularity=5, res=32, t=1, init_v_factor=0.18,
                 friction_coefficient=0, seed=None, use_colors=False, rng=None):
        super().__init__(n, r, m, hw, granularity, res, t, init_v_factor, friction_coefficient, seed, use_colors, rng)
        self.G = 0.5

    def init_x(self):
        x = None
        for _ in range(1000):
            x = self.rng.rand(self.n, 2) * 0.9 * self.hw / 2 + self.hw / 2
            if all(_row_norm(x[i] - x[j]) > self.hw / 3 for i in range(self.n) for j in range(i)):
                break
        return x

    def init_v(self, factor):
        centre = np.sum(self.x, 0) / self.n
        sign = self.rng.choice([-1, 1])
        full_v = np.zeros((self.n, 2))
        for i in range(self.n):
            d = -(centre - self.x[i])
            d = d / _row_norm(d)
            full_v[i] = np.array([sign * d[1] * (factor + 0.13 * self.rng.randn()),
                                  -sign * d[0] * (factor + 0.13 * self.rng.randn())])
        return full_v

    def step(self, action=None):
        return super().step(action, True)

    def simulate_physics(self, actions):
        mid = np.array([self.hw / 2, self.hw / 2])
        v = np.zeros_like(self.v)
        for i in range(self.n):
            force = np.array([0., 0.])
            for j in range(self.n):
                if i != j:
                    dist = np.linalg.norm(self.x[j] - self.x[i])
                    force -= self.G * self.m[j] * self.m[i] * (self.x[i] - self.x[j]) / ((dist + 1e-5) ** 3)
            to_mid = mid - self.x[i]
            force += 0.001 * (to_mid ** 3) / _row_norm(to_mid)
            force = np.clip(force, -1, 1)
            v[i] = self.v[i] + (force / self.m[i]) * self.t * self.eps
        return v


class AvoidanceTask:
    """Ball 0 is moved by one of 9 discrete actions; reward -1 whenever it touches another ball."""

    _d = 1. / np.sqrt(2)
    action_selection = [np.array(a) for a in
                        ([0., 0.], [1., 0.], [0., 1.], [_d, _d], [-1., 0.], [0., -1.], [-_d, -_d], [-_d, _d], [_d, -_d])]

    def __init__(self, env, num_stacked=4, greyscale=False, action_force=.3):
        self.env = env
        self.env.m[0] = 10000            # the controlled ball is quasi-static
        self.action_force = action_force

    def get_action_space(self):
        return len(self.action_selection)

    def step(self, action_idx):
        img, state, done = self.env.step(self.action_selection[action_idx] * self.action_force)
        return img, state, -self.env.collisions, done


class MonteCarloActionPolicy:
    """Keeps the current action with probability 1 - prob_change, else jumps uniformly."""

    def __init__(self, action_space=9, prob_change=0.1, rng=None):
        self.rng = rng if rng is not None else np.random.RandomState()
        self.action_space = action_space
        self.p = prob_change
        self.current_state = self.rng.randint(action_space)

    def next(self):
        w = self.p / (self.action_space - 1) * np.ones(self.action_space)
        w[self.current_state] = 1 - self.p
        self.current_state = self.rng.choice(range(self.action_space), p=w)
        return self.current_state


# ------------------------------------------------------------------------------------------------
# batched synthetic data in the layout the model consumes: (B, T, 3, res, res) float32 in [0, 1]
# ------------------------------------------------------------------------------------------------
ENV_PRESETS = {
    # BASELINE.json configs
    'billiards': dict(cls='billiards', n=3, r=1.2, m=1., hw=10, granularity=10, res=32, t=1., friction_coefficient=0.),
    'multibilliards': dict(cls='billiards', n=6, r=1., m=1., hw=10, granularity=10, res=32, t=1.,
                           friction_coefficient=0., use_colors=False),
    'gravity': dict(cls='gravity', n=3, r=2, m=4., hw=30, granularity=50, res=32, t=1., init_v_factor=0.55,
                    friction_coefficient=0.),
    'avoidance': dict(cls='billiards', n=3, r=1., m=1., hw=10, granularity=50, res=32, t=1., friction_coefficient=0.),
}


def make_env(preset, seed, res=None):
    """res: frame side override (the presets are BASELINE.json's 32 x 32; the reference's stock gravity / multibilliards generators
    render 50 x 50, envs.py:771-773, 841-844)."""
    cfg = dict(ENV_PRESETS[preset])
    if res is not None:
        cfg['res'] = int(res)
    cls = {'billiards': BillardsEnv, 'gravity': GravityEnv}[cfg.pop('cls')]
    return cls(seed=seed, **cfg)


def synth_sequences(preset, n_seq, t_len, seed0=0, with_actions=None, res=None):
    """One environment per sequence (seed = seed0 + i), t_len steps.
    -> dict(X (B,T,3,res,res) f32, y (B,T,n,4) f64 [, action (B,T,9), reward (B,T,1)])."""
    with_actions = (preset == 'avoidance') if with_actions is None else with_actions
    X, Y, A, R = [], [], [], []
    for i in range(n_seq):
        env = make_env(preset, seed0 + i, res)
        imgs, states = [], []
        if with_actions:
            task = AvoidanceTask(env, 4, greyscale=False, action_force=0.6)
            policy = MonteCarloActionPolicy(9, env.rng.uniform(0.2, 0.3), rng=env.rng)
            acts, rews = np.zeros((t_len, 9)), np.zeros((t_len, 1))
            for t in range(t_len):
                a = policy.next()
                img, st, rew, _ = task.step(a)
                imgs.append(img)
                states.append(st)
                acts[t - 1, a] = 1                     # stored one step back, as the reference does (envs.py:719)
                rews[t] = rew
            A.append(acts)
            R.append(rews)
        else:
            for _ in range(t_len):
                img, st, _ = env.step()
                imgs.append(img)
                states.append(st)
        X.append(np.stack(imgs))
        Y.append(np.stack(states))
    out = {'X': np.transpose(np.stack(X), (0, 1, 4, 2, 3)).astype(np.float32), 'y': np.stack(Y)}
    if with_actions:
        out['action'] = np.stack(A)
        out['reward'] = np.stack(R)
    return out


def _synth_chunk(args):
    preset, n, t_len, seed0, res, with_actions = args
    return synth_sequences(preset, n, t_len, seed0=seed0, with_actions=with_actions, res=res)


# bump when the simulators change what they draw: part of the name of every cache file of generated sequences (bench.py)
SIMULATOR_VERSION = 2


def synth_sequences_parallel(preset, n_seq, t_len, seed0=0, res=None, workers=None, with_actions=None):
    """synth_sequences over a pool of forked workers (sequence i is its own environment with seed seed0 + i, so the result
    does not depend on the split).  Fork-based: it refuses to run once the process has initialised the GPU (a forked child of
    such a process must not touch HIP, and the parent's runtime threads do not survive the fork); the workers are pure numpy."""
    import multiprocessing as mp
    import os
    import sys
    if 'torch' in sys.modules and sys.modules['torch'].cuda.is_initialized():
        raise RuntimeError('synth_sequences_parallel forks: call it before the process touches the GPU (or use synth_sequences)')
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
    workers = max(1, min(workers, n_seq))
    if workers == 1:
        return synth_sequences(preset, n_seq, t_len, seed0=seed0, with_actions=with_actions, res=res)
    per = (n_seq + workers - 1) // workers
    jobs = [(preset, min(per, n_seq - s), t_len, seed0 + s, res, with_actions) for s in range(0, n_seq, per)]
    with mp.get_context('fork').Pool(len(jobs)) as pool:
        parts = pool.map(_synth_chunk, jobs)
    return {k: np.concatenate([p[k] for p in parts], 0) for k in parts[0]}
