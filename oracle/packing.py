"""ORACLE (test infrastructure) -- the packing environment without physics.

Restates, in float64 numpy and in the reference's own order of operations:

    ItemCreator / LoadItemCreator   environment/physics0/IRcreator.py:6-24, 74-103
    PackingGame.__init__ (obs_len)  environment/physics0/binPhy.py:81-102
    PackingGame.reset               binPhy.py:128-147
    get_ratio / get_item_ratio      binPhy.py:149-156
    get_action_candidates           binPhy.py:161-169
    cur_observation                 binPhy.py:183-232
    action_to_position / prejudge   binPhy.py:234-245
    step                            binPhy.py:248-337, taking the ``simulation=False``
                                    branch (:288-289) and the ``only_simulate_current``
                                    heightmap update (:316-317)
    Interface.simulateHeight        environment/physics0/Interface.py:365-369 with the
                                    x100 scaling of Interface.py:39-40,185-187

pybullet/trimesh are reduced to kinematics: an item dropped at action cell (lx, ly)
with rotation r rests at FLB = (lx*resA, ly*resA, posZmap[r,lx,ly]) and its AABB is
FLB + extents_r.  PARITY UNPINNED for that reduction (pybullet is absent and the
reference holds no fixture for it); the Python control flow around it is pinned by
tests/golden (generated from the reference's own binPhy.py with a kinematic
Interface stand-in).

One documented deviation: ``np.argsort`` at binPhy.py:211,219 is an unstable sort
whose tie order is unspecified and differs between numpy builds; the oracle (and
the HIP path) use the stable order (ties by ascending index).
"""
from __future__ import annotations

import copy

import numpy as np

from .cvtools import getConvexHullActions
from .space import Space


class ItemCreator(object):
    """IRcreator.py:6-24."""

    def __init__(self):
        self.item_list = []

    def reset(self, index=None):
        self.item_list.clear()

    def generate_item(self, **kwargs):
        pass

    def preview(self, length):
        while len(self.item_list) < length:
            self.generate_item()
        return copy.deepcopy(self.item_list[:length])

    def update_item_queue(self, index):
        self.item_list.pop(index)


class SequenceItemCreator(ItemCreator):
    """LoadItemCreator (IRcreator.py:74-103) over ``int32[n_traj, L]`` trajectories.

    ``reset()`` advances the trajectory index *before* use (:86-92); with
    ``first_traj=1, stride=1`` the episodes read trajectories 1, 2, 3, ... exactly as
    the reference does.  A vectorised run gives bin ``g`` of ``N`` the start
    ``first_traj + g`` and stride ``N``.  Past the end of a trajectory the reference
    appends ``None`` (:97,103); here the sentinel is -1.
    """

    def __init__(self, sequences, first_traj=1, stride=1):
        super().__init__()
        self.item_trajs = np.asarray(sequences)
        self.traj_nums = len(self.item_trajs)
        self.stride = stride
        self.traj_index = first_traj - stride
        self.item_index = 0

    def reset(self, traj_index=None):
        self.item_list.clear()
        if traj_index is None:
            self.traj_index += self.stride
        else:
            self.traj_index = traj_index
        self.traj = self.item_trajs[self.traj_index % self.traj_nums]
        self.item_index = 0

    def generate_item(self, **kwargs):
        if self.item_index < len(self.traj):
            self.item_list.append(int(self.traj[self.item_index]))
        else:
            self.item_list.append(-1)
        self.item_index += 1


class RandomStreamItemCreator(ItemCreator):
    """RandomItemCreator / RandomInstanceCreator / RandomCateCreator (IRcreator.py:26-72): every item is one or two
    ``np.random.choice`` draws on the worker's generator, which envs.py:41 -> binPhy.py:118-123 seeds with
    ``seed + rank``.  The worker's global generator is a ``RandomState(seed)`` here so that several environments can
    live in one process; ``reset`` only clears the queue (IRcreator.py:11-12), the stream goes on."""

    def __init__(self, seed, dic_path=None, kind="instance", n_items=None):
        super().__init__()
        self.rs = np.random.RandomState(seed)
        self.kind = kind
        if kind == "instance":                                   # IRcreator.py:35-46
            self.groups = {}
            for k in dic_path.keys():
                self.groups.setdefault(dic_path[k][0:-6], []).append(k)
        elif kind == "category":                                 # IRcreator.py:53-68
            self.groups = {"objects": [], "concave": [], "board": []}
            for k, item in zip(dic_path.keys(), dic_path.values()):
                self.groups[item.split("/")[0]].append(k)
        else:                                                    # 'pose': RandomItemCreator over np.arange(n)
            self.item_set = np.arange(0, n_items)

    def generate_item(self, **kwargs):
        if self.kind == "pose":
            self.item_list.append(int(self.rs.choice(self.item_set)))                  # IRcreator.py:32-33
        else:
            name = self.rs.choice(list(self.groups.keys()))                             # IRcreator.py:49-51, 70-72
            self.item_list.append(int(self.rs.choice(self.groups[name])))


class PackingGame(object):
    """binPhy.py:21-337 without pybullet."""

    def __init__(self, shapes, sequences, resolutionA=0.02, resolutionH=0.01, resolutionZ=0.01,
                 bin_dimension=(0.32, 0.32, 0.30), selectedAction=500, bufferSize=1,
                 scale=(100, 100, 100), first_traj=1, traj_stride=1, stability=0, item_creator=None):
        self.stability = stability           # the stability proxy (oracle/stability.py); 0 = the reference's no-physics path
        self.last_stable = False
        self.resolutionAct = resolutionA
        self.resolutionH = resolutionH
        self.bin_dimension = np.round(np.asarray(bin_dimension, dtype=np.float64), decimals=6)  # arguments.py:115
        self.scale = np.asarray(scale, dtype=np.float64)
        self.shapes = shapes
        self.ZRotNum = shapes.n_rot
        self.selectedAction = selectedAction
        self.bufferSize = bufferSize
        self.chooseItem = self.bufferSize > 1
        self.heightResolution = resolutionZ
        self.rangeX_A, self.rangeY_A = np.ceil(self.bin_dimension[0:2] / self.resolutionAct).astype(np.int32)
        self.space = Space(self.bin_dimension, self.resolutionAct, self.resolutionH, self.ZRotNum,
                           shapes.shot_info(), shapes.extents)
        self.item_creator = item_creator if item_creator is not None else SequenceItemCreator(sequences, first_traj, traj_stride)
        self.next_item_vec = np.zeros((9))
        self.item_vec = np.zeros((1000, 9))
        self.item_idx = 0
        self.rotNum = self.ZRotNum
        self.act_len = self.bufferSize if self.chooseItem else self.selectedAction
        if not self.chooseItem:                                  # binPhy.py:87-98
            self.obs_len = len(self.next_item_vec.reshape(-1)) + self.selectedAction * 5
        else:
            self.obs_len = self.bufferSize
        self.obs_len += self.space.heightmapC.size
        # Interface.__init__: self.bin = round(bin * scale, 6)  (Interface.py:39-40)
        self.interface_bin = np.round(self.bin_dimension * self.scale, decimals=6)
        self.orderAction = 0
        self.candidates = None
        self.next_item_ID = None
        self.packed = []

    # -- episode control ---------------------------------------------------------------
    def reset(self, index=None):
        self.space.reset()
        self.item_creator.reset(index)
        self.packed = []
        self.next_item_vec[:] = 0
        self.item_idx = 0
        self.item_vec[:] = 0
        return self.cur_observation()

    def get_ratio(self):
        totalVolume = 0
        for idx in range(self.item_idx):
            totalVolume += self.shapes.volumes[int(self.item_vec[idx][0])]
        return totalVolume / np.prod(self.bin_dimension)

    def get_item_ratio(self, next_item_ID):
        return self.shapes.volumes[next_item_ID] / np.prod(self.bin_dimension)

    def gen_next_item_ID(self):
        return self.item_creator.preview(1)[0]

    def get_action_candidates(self, orderAction):
        self.next_item_ID = self.next_k_item_ID[orderAction]
        self.space.get_possible_position(self.next_item_ID)      # binPhy.py:164 (repeated at :193)
        self.chooseItem = False
        locObservation = self.cur_observation(genItem=False)
        self.chooseItem = True
        self.orderAction = orderAction
        return locObservation

    def get_all_possible_observation(self):                      # binPhy.py:171-180 (no caller in the reference)
        self.chooseItem = False
        all_obs = []
        for itemID in self.next_k_item_ID:
            self.next_item_ID = itemID
            self.space.get_possible_position(self.next_item_ID)
            all_obs.append(self.cur_observation(genItem=False))
        self.chooseItem = True       # (the reference leaves it False, which would turn the next order observation into a location
        return np.concatenate(all_obs, axis=0)      # observation with a fresh item: nobody calls it there; kept usable here)

    # -- observation -------------------------------------------------------------------
    def cur_observation(self, genItem=True):
        if not self.chooseItem:
            if genItem:
                self.next_item_ID = self.gen_next_item_ID()
            self.next_item_vec[0] = self.next_item_ID
            self.space.get_possible_position(self.next_item_ID)
            result = np.concatenate((self.next_item_vec.reshape(-1), self.space.heightmapC.reshape(-1)))
            self.candidates = getConvexHullActions(self.space.posZValid, self.space.naiveMask,
                                                   self.heightResolution)
            if self.candidates is not None:
                if len(self.candidates) > self.selectedAction:
                    selectedIndex = np.argsort(self.candidates[:, 3], kind='stable')[0: self.selectedAction]
                    self.candidates = self.candidates[selectedIndex]
                elif len(self.candidates) < self.selectedAction:
                    dif = self.selectedAction - len(self.candidates)
                    self.candidates = np.concatenate((self.candidates, np.zeros((dif, 5))), axis=0)
            if self.candidates is None:
                poszFlatten = self.space.posZValid.reshape(-1)
                selectedIndex = np.argsort(poszFlatten, kind='stable')[0: self.selectedAction]
                ROT, X, Y = np.unravel_index(selectedIndex, (self.rotNum, self.rangeX_A, self.rangeY_A))
                H = poszFlatten[selectedIndex]
                V = self.space.naiveMask.reshape(-1)[selectedIndex]
                H[:] = self.bin_dimension[-1]
                self.candidates = np.concatenate([ROT.reshape(-1, 1), X.reshape(-1, 1), Y.reshape(-1, 1),
                                                  H.reshape(-1, 1), V.reshape(-1, 1)], axis=1)
                if len(self.candidates) < self.selectedAction:   # R*Ax*Ay < S never pads in the reference;
                    dif = self.selectedAction - len(self.candidates)   # keep the obs length fixed
                    self.candidates = np.concatenate((self.candidates, np.zeros((dif, 5))), axis=0)
            result = np.concatenate((self.candidates.reshape(-1), result))
        else:
            self.next_k_item_ID = self.item_creator.preview(self.bufferSize)
            result = np.concatenate((np.array(self.next_k_item_ID), self.space.heightmapC.reshape(-1)))
        return result

    # -- action ------------------------------------------------------------------------
    def action_to_position(self, action):
        rotIdx, lx, ly = self.candidates[action][0:3].astype(int)
        return rotIdx, np.round((lx * self.resolutionAct, ly * self.resolutionAct, self.bin_dimension[2]),
                                decimals=6), (lx, ly)

    def prejudge(self, rotIdx, translation, naiveMask):
        if self.next_item_ID is None or self.next_item_ID < 0:
            return False                                          # exhausted trajectory (sentinel)
        extents = self.shapes.extents[self.next_item_ID][rotIdx]
        if np.round(translation[0] + extents[0] - self.bin_dimension[0], decimals=6) > 0 \
                or np.round(translation[1] + extents[1] - self.bin_dimension[1], decimals=6) > 0:
            return False
        if np.sum(naiveMask) == 0:
            return False
        return True

    def simulateHeight(self, rotIdx, height):
        """Interface.simulateHeight (Interface.py:365-369) on the kinematic AABB, x100 units."""
        extents = self.shapes.extents[self.next_item_ID][rotIdx]
        maxC_z = height * self.scale[2] + extents[2] * self.scale[2]
        if np.round(maxC_z - self.interface_bin[2], decimals=6) > 0:
            return False, True
        return True, True

    def step(self, action):
        rotIdx, targetFLB, coordinate = self.action_to_position(action)
        success = self.prejudge(rotIdx, targetFLB, self.space.naiveMask)
        height = self.space.posZmap[rotIdx, coordinate[0], coordinate[1]]
        if success:
            success, sim_suc = self.simulateHeight(rotIdx, height)
        self.last_stable = False
        if success and self.stability:
            from .stability import placement_is_stable
            T, B, maskH, maskB = self.shapes.tables[self.next_item_ID][rotIdx]
            X, Y = coordinate[0] * self.space.stepSize, coordinate[1] * self.space.stepSize
            window = self.space.heightmapC[X:X + T.shape[0], Y:Y + T.shape[1]]
            self.last_stable = placement_is_stable(window, T, B, maskH, maskB, height, self.heightResolution)
            if self.stability == 2 and not self.last_stable:
                success = False
        self.packed.append([self.next_item_ID, int(rotIdx), int(coordinate[0]), int(coordinate[1]), float(height)])

        if not success:
            reward = 0.0
            info = {'counter': self.item_idx, 'ratio': self.get_ratio(), 'Valid': True}
            observation = self.cur_observation()
            return observation, reward, True, info

        self.space.place_item(self.next_item_ID, rotIdx, coordinate[0], coordinate[1], height)
        self.item_vec[self.item_idx, 0] = self.next_item_ID
        self.item_vec[self.item_idx, -1] = 1
        item_ratio = self.get_item_ratio(self.next_item_ID)
        reward = item_ratio * 10
        self.item_idx += 1
        self.item_creator.update_item_queue(self.orderAction)
        self.item_creator.generate_item()
        observation = self.cur_observation()
        return observation, reward, False, {'Valid': True}


class OracleVecEnv(object):
    """N sequential PackingGame instances behind the ShmemVecEnv protocol:
    auto-reset on done (wrapper/shmem_vec_env.py:141-144) and the Monitor's
    ``info['episode']`` (wrapper/monitor.py:58-75).  Observations stay float64;
    ``to_float32`` applies the VecPyTorch cast (envs.py:151,163)."""

    def __init__(self, num_envs, shapes, sequences, traj_start=1, global_offset=0, global_num=None, item_creators=None, **kw):
        global_num = num_envs if global_num is None else global_num
        self.envs = [PackingGame(shapes, sequences, first_traj=traj_start + global_offset + g, traj_stride=global_num,
                                 item_creator=None if item_creators is None else item_creators[g], **kw)
                     for g in range(num_envs)]
        self.num_envs = num_envs
        self.obs_len = self.envs[0].obs_len
        self.rewards = [[] for _ in range(num_envs)]

    def reset(self):
        self.rewards = [[] for _ in range(self.num_envs)]
        return np.array([e.reset() for e in self.envs])

    def reset_specific(self, indexs):
        """shmem_vec_env.py:113-117: ``env.reset()`` of the listed workers; Monitor.reset drops the
        rewards of the abandoned episode (monitor.py:47-56)."""
        for i in indexs:
            self.rewards[i] = []
        return np.array([self.envs[i].reset() for i in indexs])

    def get_action_candidates(self, order_actions):
        return np.array([e.get_action_candidates(int(a)) for e, a in zip(self.envs, order_actions)])

    def get_all_possible_observation(self):
        return np.array([e.get_all_possible_observation() for e in self.envs])

    def step(self, actions):
        obs, rews, dones, infos = [], [], [], []
        for i, (e, a) in enumerate(zip(self.envs, actions)):
            o, r, d, info = e.step(int(a))
            self.rewards[i].append(r)
            if d:
                eprew = sum(self.rewards[i])
                info['episode'] = {'r': round(eprew, 6), 'l': len(self.rewards[i])}
                self.rewards[i] = []
                o = e.reset()
            obs.append(o)
            rews.append(r)
            dones.append(d)
            infos.append(info)
        return np.array(obs), np.array(rews), np.array(dones), infos

    @staticmethod
    def to_float32(obs):
        return np.asarray(obs, dtype=np.float64).astype(np.float32)
