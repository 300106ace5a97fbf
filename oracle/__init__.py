"""oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/carenv_oracle.c (the CPU restatement of the reference hot path)
plus a restatement of CarEnv.load_track.  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this package; ppo-car_amd/ never does.
Parity status: pinned against tests/golden/*.npz (see carenv_oracle.c header).
"""
import ctypes as C
import json
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    """Compile liboracle.so (gcc, seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "carenv_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oc_ray_distance.restype = C.c_double
        L.oc_ray_distance.argtypes = [C.c_double, C.c_double, C.c_double, _f64p, C.c_int]
        L.oc_ray_count.restype = C.c_int
        L.oc_ray_count.argtypes = [C.c_int]
        L.oc_set_norm_mode.argtypes = [C.c_int]
        st = [C.c_void_p] * 9  # px py vx vy rot time_step next_gate passed destroyed (raw pointers: sliced views)
        L.oc_env_step.restype = None
        L.oc_env_step.argtypes = [_f64p, C.c_int, _f64p, C.c_int, C.c_int, C.c_int64] + st + [C.c_void_p] * 5
        L.oc_env_reset.restype = None
        L.oc_env_reset.argtypes = [_f64p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_int64] + st + [C.c_void_p]
        L.oc_vec_step.restype = None
        L.oc_vec_step.argtypes = ([_f64p, C.c_int, _f64p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_int64] + st + [C.c_void_p] * 6)
        L.oc_vec_rollout_mt.restype = C.c_double
        L.oc_vec_rollout_mt.argtypes = ([_f64p, C.c_int, _f64p, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                         C.c_int64, C.c_int64, C.c_int] + st + [_i64p])
        L.oc_gae.restype = None
        L.oc_gae.argtypes = [_f32p] * 7 + [C.c_double, C.c_double, C.c_int64, C.c_int64, _f32p, _f32p]
        _LIB = L
    return _LIB


class Track:
    """CarEnv.load_track (car_env.py:535-567) + the geometry lists CarEnv.reset builds
    (car_env.py:651-676): walls = outer segments then inner segments, gates = consecutive
    point pairs, everything scaled x*1280, y*720."""

    def __init__(self, path):
        with open(path, "r") as f:
            data = json.load(f)
        W, H = 1280, 720
        outer = [[x * W, y * H] for x, y in data["outer_track_points"]]
        inner = [[x * W, y * H] for x, y in data["inner_track_points"]]
        gpts = [[x * W, y * H] for x, y in data["reward_gates"]]
        walls = [outer[b] + outer[b + 1] for b in range(len(outer) - 1)]
        walls += [inner[b] + inner[b + 1] for b in range(len(inner) - 1)]
        gates = [a + b for a, b in zip(gpts[::2], gpts[1::2])]
        self.walls = np.ascontiguousarray(np.array(walls, np.float64).reshape(-1, 4))
        self.gates = np.ascontiguousarray(np.array(gates, np.float64).reshape(-1, 4))
        self.S, self.G = len(walls), len(gates)
        self.start_x = data["initial_position"][0] * W
        self.start_y = data["initial_position"][1] * H
        self.start_rot = float(data["initial_angle"])


def ray_count(n):
    return lib().oc_ray_count(n)


class OracleVecEnv:
    """n_envs independent oracle environments on one track, SoA float64 state."""

    FIELDS = ("px", "py", "vx", "vy", "rot", "time_step", "next_gate", "passed", "destroyed")

    def __init__(self, track, n_envs, num_rays=12, reward_scaling=1.0, threads=1):
        self.t, self.N, self.n = track, int(n_envs), int(num_rays)
        self.R = ray_count(num_rays)
        self.D = 6 + self.R
        self.reward_scaling = float(reward_scaling)
        self.threads = int(threads)
        self.px, self.py, self.vx, self.vy, self.rot = (np.zeros(self.N, np.float64) for _ in range(5))
        self.time_step, self.next_gate, self.passed = (np.zeros(self.N, np.int64) for _ in range(3))
        self.destroyed = np.zeros(self.N, np.uint8)
        self._pool = ThreadPoolExecutor(self.threads) if self.threads > 1 else None

    def _state_ptrs(self, lo):
        return [getattr(self, f)[lo:].ctypes.data for f in self.FIELDS]

    def _chunks(self):
        k = self.threads
        b = [self.N * i // k for i in range(k + 1)]
        return [(b[i], b[i + 1]) for i in range(k) if b[i + 1] > b[i]]

    def _run(self, fn):
        ch = self._chunks()
        if len(ch) == 1:
            fn(*ch[0])
        else:
            list(self._pool.map(lambda c: fn(*c), ch))

    def reset(self):
        obs = np.zeros((self.N, self.D), np.float32)
        t = self.t
        lib().oc_env_reset(t.walls, t.S, self.n, t.start_x, t.start_y, t.start_rot, self.N,
                           *self._state_ptrs(0), obs.ctypes.data)
        return obs

    def set_state(self, **kw):
        for k, v in kw.items():
            getattr(self, k)[:] = v

    def raw_step(self, action):
        """CarEnv.step without auto-reset / reward scaling -> obs, reward(f64), term, trunc."""
        action = np.ascontiguousarray(action, np.int64)
        obs = np.zeros((self.N, self.D), np.float32)
        rew = np.zeros(self.N, np.float64)
        term, trunc = np.zeros(self.N, np.uint8), np.zeros(self.N, np.uint8)
        t = self.t

        def fn(lo, hi):
            lib().oc_env_step(t.walls, t.S, t.gates, t.G, self.n, hi - lo, *self._state_ptrs(lo),
                              action[lo:].ctypes.data, obs[lo:].ctypes.data, rew[lo:].ctypes.data,
                              term[lo:].ctypes.data, trunc[lo:].ctypes.data)
        self._run(fn)
        return obs, rew, term.astype(bool), trunc.astype(bool)

    def step(self, action, want_final_obs=False):
        """The vector-env call of train.py:185 (auto-reset, reward * reward_scaling)."""
        action = np.ascontiguousarray(action, np.int64)
        obs = np.zeros((self.N, self.D), np.float32)
        fin = np.zeros((self.N, self.D), np.float32) if want_final_obs else None
        rew = np.zeros(self.N, np.float64)
        term, trunc = np.zeros(self.N, np.uint8), np.zeros(self.N, np.uint8)
        t = self.t

        def fn(lo, hi):
            lib().oc_vec_step(t.walls, t.S, t.gates, t.G, self.n, t.start_x, t.start_y, t.start_rot,
                              self.reward_scaling, hi - lo, *self._state_ptrs(lo), action[lo:].ctypes.data,
                              obs[lo:].ctypes.data, rew[lo:].ctypes.data, term[lo:].ctypes.data,
                              trunc[lo:].ctypes.data, fin[lo:].ctypes.data if fin is not None else None)
        self._run(fn)
        out = (obs, rew, term.astype(bool), trunc.astype(bool))
        return out + (fin,) if want_final_obs else out


    def run_steps(self, actions, threads=None):
        """bench.py's CPU baseline: T vector-env steps (auto-reset, reward scaling) with pre-generated actions [T, N] inside ONE C call,
        the envs dealt to `threads` POSIX threads in static contiguous ranges (oc_vec_rollout_mt).  Returns the sum of the rewards."""
        actions = np.ascontiguousarray(actions, np.int64)
        T, N = actions.shape
        assert N == self.N
        t = self.t
        return lib().oc_vec_rollout_mt(t.walls, t.S, t.gates, t.G, self.n, t.start_x, t.start_y, t.start_rot, self.reward_scaling, N, T,
                                       int(threads or self.threads), *self._state_ptrs(0), actions)


def ray_distance(px, py, angle_deg, segs):
    segs = np.ascontiguousarray(np.asarray(segs, np.float64).reshape(-1, 4))
    return lib().oc_ray_distance(px, py, angle_deg, segs, len(segs))


def gae(rew, val, term, trunc, last_val, last_term, last_trunc, gamma=0.99, lam=0.95):
    f = lambda a: np.ascontiguousarray(a, np.float32)
    rew, val, term, trunc = f(rew), f(val), f(term), f(trunc)
    T, N = rew.shape
    adv, ret = np.zeros((T, N), np.float32), np.zeros((T, N), np.float32)
    lib().oc_gae(rew, val, term, trunc, f(last_val).reshape(-1), f(last_term).reshape(-1), f(last_trunc).reshape(-1),
                 gamma, lam, T, N, adv, ret)
    return adv, ret
