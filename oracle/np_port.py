"""ORACLE (tier 1) -- NumPy/Python restatement of the reference's env step IN THE REFERENCE'S
SHAPE: one env per object, a Python loop over `update_frequency` sim iterations, the
controller every 15th, NumPy float64 on 3xN arrays.  Test infrastructure only; it is also
what bench.py times as `cpu_baseline` (kind "port": the reference itself cannot travel to
the GPU box and its simulator, rps, is absent everywhere).

Follows, layer by layer (paths relative to /root/reference/robotarium_gym/):
  utilities/roboEnv.py:11-121            -> RoboEnvPort
  utilities/controller.py:4-24           -> ControllerPort
  utilities/misc.py:20-25,49-63          -> nearest_neighbors, generate_initial_locations
  scenarios/PredatorCapturePrey/{PredatorCapturePrey.py:14-222, agent.py:4-76} -> PCPPort
  scenarios/Warehouse/warehouse.py:10-195                                     -> WarehousePort
  scenarios/MaterialTransport/MaterialTransport.py:10-207                     -> MTPort
over the restated rps in oracle/rps_restated (parity vs real rps + cvxopt unpinned).
tests/test_oracle_golden.py checks every class against tests/golden/ (captured from the
reference's own Python) for float64 equality.
"""
import copy
import os
import sys

import numpy as np

_RPS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rps_restated")
if _RPS not in sys.path:
    sys.path.insert(0, _RPS)

import rps.robotarium as robotarium  # noqa: E402
from rps.utilities.barrier_certificates import (  # noqa: E402
    create_si_to_uni_mapping, create_single_integrator_barrier_certificate,
    create_single_integrator_barrier_certificate2)
from rps.utilities.controllers import create_si_position_controller  # noqa: E402
from rps.utilities.misc import generate_initial_conditions  # noqa: E402


class Args(object):
    def __init__(self, d):
        self.__dict__ = dict(d)


class ControllerPort(object):  # utilities/controller.py:4-24
    def __init__(self, type='safe', family=None):
        self.position_controller = create_si_position_controller()
        self.si_to_uni_dyn, self.uni_to_si_states = create_si_to_uni_mapping()
        if family:   # Controller('custom', <one of rps' factories with other arguments>): config keys of the certificate family
            kw = {"barrier_gain": family.get("barrier_gain", 100), "magnitude_limit": family.get("magnitude_limit", 0.2)}
            if type == "safe":
                self.si_barrier_cert = create_single_integrator_barrier_certificate2(
                    unsafe_barrier_gain=family.get("unsafe_barrier_gain", 1e6), safety_radius=family.get("safety_radius", 0.2), **kw)
            else:
                self.si_barrier_cert = create_single_integrator_barrier_certificate(safety_radius=family.get("safety_radius", 0.17), **kw)
        elif type == "safe":
            self.si_barrier_cert = create_single_integrator_barrier_certificate2(safety_radius=.2)
        elif type == "default":
            self.si_barrier_cert = create_single_integrator_barrier_certificate()
        else:
            raise ValueError(type)

    def set_velocities(self, agent_poses, goals):
        xi = self.uni_to_si_states(agent_poses)
        dxi = self.position_controller(xi, goals[:2][:])
        dxi = self.si_barrier_cert(dxi, xi)
        return self.si_to_uni_dyn(dxi, agent_poses)


class RoboEnvPort(object):  # utilities/roboEnv.py:11-121
    def __init__(self, agents, args):
        self.args = args
        self.agents = agents
        family = {k: getattr(args, k) for k in ("safety_radius", "barrier_gain", "unsafe_barrier_gain", "magnitude_limit") if hasattr(args, k)}
        self.controller = ControllerPort(getattr(args, "barrier_certificate", "safe"), family)
        self.first_run = True
        self.errors = {}
        self.previous_pose = None

    def reset(self):
        self.robotarium = robotarium.Robotarium(number_of_robots=self.agents.num_robots, show_figure=False,
                                                initial_conditions=self.agents.agent_poses,
                                                sim_in_real_time=False)
        self.agents.agent_poses = self.robotarium.get_poses()
        self.robotarium.step()
        self.previous_pose = None

    def step(self, actions_):
        goals_ = self.agents._generate_step_goal_positions(actions_)
        dist_travelled = np.zeros((self.agents.num_robots))
        message = ''
        for iterations in range(self.args.update_frequency):
            self.agents.agent_poses = self.robotarium.get_poses()
            if self.previous_pose is not None:
                dist_travelled += np.linalg.norm(self.agents.agent_poses[:2, :] - self.previous_pose[:2, :], axis=0)
            self.previous_pose = copy.deepcopy(self.agents.agent_poses)
            if iterations % 15 == 0 or self.args.robotarium:
                velocities = self.controller.set_velocities(self.agents.agent_poses, goals_)
                self.robotarium.set_velocities(np.arange(self.agents.num_robots), velocities)
            self.robotarium.step()
            if self.args.penalize_violations:
                errs = self.robotarium._errors
                if 'collision' in errs and ('collision' not in self.errors or
                                            sum(errs['collision'].values()) > sum(self.errors['collision'].values())):
                    message = 'collision'
                if 'boundary' in errs and ('boundary' not in self.errors or
                                           sum(errs['boundary'].values()) > sum(self.errors['boundary'].values())):
                    message = 'boundary' if message == '' else message + "_boundary"
                self.errors = copy.deepcopy(errs)
                if message != '':
                    dist_travelled += np.linalg.norm(self.agents.agent_poses[:2, :] - self.previous_pose[:2, :], axis=0)
                    return message, dist_travelled
        return "", dist_travelled


def nearest_neighbors(poses, agent, num_neighbors):
    """misc.py:20-25 with the canonical order (ascending distance, ties -> lower index) in place
    of np.argpartition's implementation-defined one."""
    N = poses.shape[1]
    dists = [np.linalg.norm(poses[:2, x] - poses[:2, agent]) for x in range(N)]
    order = sorted((j for j in range(N) if j != agent), key=lambda j: (dists[j], j))
    return order[:num_neighbors]


def generate_initial_locations(num_locs, width, height, thresh, start_dist=.3, spawn_left=True):  # misc.py:49-63
    poses = generate_initial_conditions(num_locs, spacing=start_dist, width=width, height=height)
    for i in range(len(poses[0])):
        if spawn_left:
            poses[0][i] -= (width / 2 - thresh)
        else:
            poses[0][i] += (width / 2 - thresh)
        poses[2][i] = 0
    return poses


def _goal(goal_pose, word, step, args):  # agent.py:48-76 / warehouse.py:19-45 / MaterialTransport.py:19-46
    def cx(v):
        return args.LEFT if v < args.LEFT else args.RIGHT if v > args.RIGHT else v

    def cy(v):
        return args.UP if v < args.UP else args.DOWN if v > args.DOWN else v
    if word == 0:
        goal_pose[0] = max(goal_pose[0] - step, args.LEFT)
        goal_pose[1] = cy(goal_pose[1])
    elif word == 1:
        goal_pose[0] = min(goal_pose[0] + step, args.RIGHT)
        goal_pose[1] = cy(goal_pose[1])
    elif word == 2:
        goal_pose[0] = cx(goal_pose[0])
        goal_pose[1] = max(goal_pose[1] - step, args.UP)
    elif word == 3:
        goal_pose[0] = cx(goal_pose[0])
        goal_pose[1] = min(goal_pose[1] + step, args.DOWN)
    else:
        goal_pose[0] = cx(goal_pose[0])
        goal_pose[1] = cy(goal_pose[1])
    return goal_pose


class _ScenarioPort(object):
    def _neighbour_obs(self, observations):
        full = []
        N, K = self.num_robots, self.args.num_neighbors
        for i in range(N):
            o = np.asarray(observations[i], dtype=np.float64)
            if K >= N - 1:
                nbr = [j for j in range(N) if j != i]
            else:
                nbr = nearest_neighbors(self.agent_poses, i, K)
            for j in nbr:
                o = np.concatenate((o, np.asarray(observations[j], dtype=np.float64)))
            full.append(o)
        return full


class PCPPort(_ScenarioPort):  # scenarios/PredatorCapturePrey/PredatorCapturePrey.py
    def __init__(self, args):
        self.args = args
        self.num_robots = args.predator + args.capture
        self.num_prey = args.num_prey
        if args.seed != -1:
            np.random.seed(args.seed)
        self.sensing = [args.predator_radius] * args.predator + [0] * args.capture
        self.capture = [0] * args.predator + [args.capture_radius] * args.capture
        self.agent_obs_dim = 6 if args.capability_aware else 4
        self.env = RoboEnvPort(self, args)

    def _generate_step_goal_positions(self, actions):
        goal = copy.deepcopy(self.agent_poses)
        for i in range(self.num_robots):
            goal[:, i] = _goal(goal[:, i], actions[i], self.args.step_dist, self.args)
        return goal

    def reset(self):
        a = self.args
        self.episode_steps = 0
        width = a.ROBOT_INIT_RIGHT_THRESH - a.LEFT
        height = a.DOWN - a.UP
        self.agent_poses = generate_initial_locations(self.num_robots, width, height, a.ROBOT_INIT_RIGHT_THRESH,
                                                      start_dist=a.start_dist)
        width = a.RIGHT - a.PREY_INIT_LEFT_THRESH
        self.prey_loc = generate_initial_locations(self.num_prey, width, height, a.ROBOT_INIT_RIGHT_THRESH,
                                                   start_dist=a.step_dist, spawn_left=False)[:2].T
        self.prey_captured = [False] * self.num_prey
        self.prey_sensed = [False] * self.num_prey
        self.prev_counts = (self.num_prey, self.num_prey)
        self.env.reset()
        return [[0] * (self.agent_obs_dim * (a.num_neighbors + 1))] * self.num_robots

    def step(self, actions_):
        a = self.args
        info = {}
        terminated = False
        self.episode_steps += 1
        message, dist = self.env.step(actions_)
        for i, prey in enumerate(self.prey_loc):  # :72-95
            if self.prey_captured[i]:
                continue
            if not self.prey_sensed[i]:
                for k in range(self.num_robots):
                    if np.linalg.norm(self.agent_poses[:2, k] - prey) <= self.sensing[k]:
                        self.prey_sensed[i] = True
                        break
            if self.prey_sensed[i]:
                for k, action in enumerate(actions_):
                    if action == 4 and np.linalg.norm(self.agent_poses[:2, k] - prey) <= self.capture[k]:
                        self.prey_captured[i] = True
                        break
        num_prey = self.num_prey - sum(self.prey_captured)
        unseen = self.num_prey - sum(self.prey_sensed)
        own = []
        for k in range(self.num_robots):  # agent.py:19-46
            closest, loc = -1, [-5, -5]
            for i in range(self.num_prey):
                if self.prey_captured[i]:
                    continue
                d = np.linalg.norm(self.agent_poses[:2, k] - self.prey_loc[i])
                if d <= self.sensing[k] and (d < closest or closest == -1):
                    loc, closest = self.prey_loc[i], d
            o = [*self.agent_poses[:2, k], *loc]
            if a.capability_aware:
                o += [self.sensing[k], self.capture[k]]
            own.append(np.array(o, dtype=np.float64))
        obs = self._neighbour_obs(own)
        if message != '':
            info['message'] = message
            terminated = True
            rewards = -5
        else:
            rewards = 0
            rewards += (self.prev_counts[1] - unseen) * a.sense_reward
            rewards += (self.prev_counts[0] - num_prey) * a.capture_reward
            rewards += a.time_penalty
            self.prev_counts = (num_prey, unseen)
            if self.episode_steps > a.max_episode_steps or num_prey == 0:
                info['remaining'] = num_prey
                terminated = True
        info['dist_travelled'] = dist
        return obs, [rewards] * self.num_robots, [terminated] * self.num_robots, info


class WarehousePort(_ScenarioPort):  # scenarios/Warehouse/warehouse.py
    def __init__(self, args):
        self.args = args
        self.num_robots = args.n_agents
        if args.seed != -1:
            np.random.seed(args.seed)
        self.green = [i % 2 == 0 for i in range(self.num_robots)]
        self.loaded = [False] * self.num_robots
        self.env = RoboEnvPort(self, args)

    def _generate_step_goal_positions(self, actions):
        goal = copy.deepcopy(self.agent_poses)
        for i in range(self.num_robots):
            goal[:, i] = _goal(goal[:, i], actions[i], self.args.step_dist, self.args)
        return goal

    def reset(self):
        a = self.args
        self.episode_steps = 0
        self.loaded = [False] * self.num_robots
        self.agent_poses = generate_initial_conditions(self.num_robots, spacing=a.start_dist,
                                                       width=a.RIGHT - a.LEFT, height=a.DOWN - a.UP)
        self.agent_poses[0] += (1.5 + a.LEFT) / 2
        self.agent_poses[0] -= (1.5 - a.RIGHT) / 2
        self.agent_poses[1] -= (1 + a.UP) / 2
        self.agent_poses[1] += (1 - a.DOWN) / 2
        self.env.reset()
        return [[0] * (3 * (a.num_neighbors + 1))] * self.num_robots

    def step(self, actions_):
        a = self.args
        self.episode_steps += 1
        info = {}
        message, dist = self.env.step(actions_)
        own = [[*self.agent_poses[:2, k], self.loaded[k]] for k in range(self.num_robots)]
        obs = self._neighbour_obs(own)
        if message == '':
            rewards = []
            for k in range(self.num_robots):  # :145-178
                pos = self.agent_poses[:2, k]
                r = 0
                if self.loaded[k]:
                    if pos[0] < -1.5 + a.goal_width and ((self.green[k] and pos[1] > 0) or
                                                         (not self.green[k] and pos[1] <= 0)):
                        r = a.unload_reward
                        self.loaded[k] = False
                else:
                    if pos[0] > 1.5 - a.goal_width and ((not self.green[k] and pos[1] > 0) or
                                                        (self.green[k] and pos[1] <= 0)):
                        r = a.load_reward
                        self.loaded[k] = True
                rewards.append(r)
            terminated = self.episode_steps > a.max_episode_steps
        else:
            info['message'] = message
            rewards = [-5] * self.num_robots
            terminated = True
        info['dist_travelled'] = dist
        return obs, rewards, [terminated] * self.num_robots, info


class MTPort(_ScenarioPort):  # scenarios/MaterialTransport/MaterialTransport.py
    def __init__(self, args):
        self.args = args
        self.num_robots = args.n_agents
        if args.seed != -1:
            np.random.seed(args.seed)
        nf = args.n_fast_agents
        self.torque = [args.small_torque] * nf + [args.large_torque] * args.n_slow_agents
        self.speed = [args.fast_step] * nf + [args.slow_step] * args.n_slow_agents
        self.agent_obs_dim = 11 if args.capability_aware else 9
        self.env = RoboEnvPort(self, args)

    def _generate_step_goal_positions(self, actions):
        goal = copy.deepcopy(self.agent_poses)
        for i in range(self.num_robots):
            goal[:, i] = _goal(goal[:, i], actions[i] // 4, self.speed[i], self.args)
        return goal

    def reset(self):
        a = self.args
        self.episode_steps = 0
        self.messages = [0, 0, 0, 0]
        z1 = {k: v for k, v in a.zone1.items() if k != 'distribution'}
        z2 = {k: v for k, v in a.zone2.items() if k != 'distribution'}
        self.zone1_load = int(getattr(np.random, a.zone1['distribution'])(**z1))
        self.zone2_load = int(getattr(np.random, a.zone2['distribution'])(**z2))
        self.load = [0] * self.num_robots
        self.agent_poses = generate_initial_locations(self.num_robots, a.end_goal_width, a.DOWN - a.UP,
                                                      a.LEFT + a.end_goal_width, start_dist=a.start_dist)
        self.env.reset()
        return [[0] * self.agent_obs_dim] * self.num_robots

    def step(self, actions_):
        a = self.args
        self.episode_steps += 1
        info = {}
        message, dist = self.env.step(actions_)
        for i in range(len(self.messages)):
            self.messages[i] = actions_[i] % 4
        obs = []
        for k in range(self.num_robots):
            o = [*self.agent_poses[:2, k], self.load[k], self.zone1_load, self.zone2_load, *self.messages]
            if a.capability_aware:
                o += [self.torque[k], self.speed[k]]
            obs.append(np.array(o, dtype=np.float64))
        if message == '':
            reward = a.time_penalty
            for k in range(self.num_robots):  # :161-189
                pos = self.agent_poses[:2, k]
                if self.load[k] > 0:
                    if pos[0] < -1.5 + a.end_goal_width:
                        reward += self.load[k] * a.unload_multiplier
                        self.load[k] = 0
                else:
                    if pos[0] > 1.5 - a.end_goal_width:
                        if self.zone2_load > self.torque[k]:
                            self.load[k] = self.torque[k]
                            self.zone2_load -= self.torque[k]
                        else:
                            self.load[k] = self.zone2_load
                            self.zone2_load = 0
                        reward += self.load[k] * a.load_multiplier
                    elif np.linalg.norm(self.agent_poses[:2, k] - [0, 0]) <= a.zone1_radius:
                        if self.zone1_load > self.torque[k]:
                            self.load[k] = self.torque[k]
                            self.zone1_load -= self.torque[k]
                        else:
                            self.load[k] = self.zone1_load
                            self.zone1_load = 0
                        reward += self.load[k] * a.load_multiplier
            terminated = self.episode_steps > a.max_episode_steps
            if not terminated:
                terminated = self.zone1_load == 0 and self.zone2_load == 0 and all(l == 0 for l in self.load)
        else:
            info['message'] = message
            reward = -6
            terminated = True
        info['dist_travelled'] = dist
        if terminated:
            info['remaining'] = self.zone1_load + self.zone2_load + sum(self.load)
        return obs, [reward] * self.num_robots, [terminated] * self.num_robots, info


class SimplePort(_ScenarioPort):  # scenarios/Simple/simple.py
    def __init__(self, args):
        self.args = args
        self.num_robots = args.n_agents
        if args.seed != -1:
            np.random.seed(args.seed)
        self.obs_dim = 2 * (self.num_robots + 1)
        self.env = RoboEnvPort(self, args)

    def _generate_step_goal_positions(self, actions):
        goal = copy.deepcopy(self.agent_poses)
        for i in range(self.num_robots):
            goal[:, i] = _goal(goal[:, i], actions[i], self.args.step_dist, self.args)
        return goal

    def reset(self):
        a = self.args
        self.episode_steps = 0
        width = a.ROBOT_INIT_RIGHT_THRESH - a.LEFT
        height = a.DOWN - a.UP
        self.agent_poses = generate_initial_locations(self.num_robots, width, height, a.ROBOT_INIT_RIGHT_THRESH,
                                                      start_dist=a.start_dist)
        width = a.RIGHT - a.PREY_INIT_LEFT_THRESH
        self.goal_loc = generate_initial_locations(1, width, height, a.ROBOT_INIT_RIGHT_THRESH,
                                                   start_dist=a.step_dist, spawn_left=False)[:2].T
        self.env.reset()
        return [[0] * self.obs_dim] * self.num_robots

    def step(self, actions_):
        a = self.args
        self.episode_steps += 1
        info = {}
        message, dist = self.env.step(actions_)
        own = [np.array(self.agent_poses[:2, k]) for k in range(self.num_robots)]
        obs = []
        for k in range(self.num_robots):
            o = own[k]
            for j in range(self.num_robots):
                if j != k:
                    o = np.concatenate((o, own[j]))
            obs.append(np.concatenate((o, self.goal_loc.reshape(-1))))
        if message == '':
            rewards = []
            for k in range(self.num_robots):
                r = -(np.sum(np.square(self.agent_poses[:2, k].reshape(1, 2) - self.goal_loc.reshape(1, 2))))
                r *= a.reward_scaler
                rewards.append(r)
            terminated = self.episode_steps > a.max_episode_steps
        else:
            rewards = [-5] * self.num_robots
            terminated = True
            info['message'] = message       # the reference files it under 'remaining' (simple.py:176)
        info['dist_travelled'] = dist
        return obs, rewards, [terminated] * self.num_robots, info


class ArcticPort(_ScenarioPort):  # scenarios/ArcticTransport/{ArcticTransport.py, agent.py}
    TYPES = ('drone', 'drone', 'ice', 'water')
    OTHERS = ((1, 2, 3), (0, 2, 3), (3, 0, 1), (2, 0, 1))

    def __init__(self, args):
        import math
        self.args = args
        self.num_robots = args.n_agents
        assert self.num_robots == 4
        self.start_poses = np.array([[-.3, .3, -.9, .9], [-.8] * 4, [math.pi / 2] * 4])
        if args.seed != -1:
            np.random.seed(args.seed)
        self.env = RoboEnvPort(self, args)

    def _step_dist(self, k):
        a, pix, t = self.args, self.pixel_type[k], self.TYPES[k]
        if t == 'drone':
            return a.fast_step
        if t == 'water':
            return a.slow_step if pix == 1 else a.fast_step if pix == 2 else a.normal_step
        return a.fast_step if pix == 1 else a.slow_step if pix == 2 else a.normal_step

    def _generate_step_goal_positions(self, actions):
        goal = copy.deepcopy(self.agent_poses)
        for i in range(self.num_robots):
            goal[:, i] = _goal(goal[:, i], actions[i], self._step_dist(i), self.args)
        return goal

    def reset(self):
        import random
        self.episode_steps = 0
        self.pixel_type = [0] * 4
        self.reached_goal = [False] * 4
        self.agent_poses = copy.deepcopy(self.start_poses)
        self.grid = np.random.randint(3, size=(8, 12))
        c = random.randint(1, 11)
        self.grid[0][c] = self.grid[0][c - 1] = self.grid[1][c] = self.grid[1][c - 1] = 3
        self.grid[7][1:11] = np.array([0] * 10)
        self.goal_loc = [1, c]
        self.env.reset()
        return [[0] * 30] * 4

    @staticmethod
    def pose_from_cell(cell):
        return [cell[1] * .25 - 1.5, (-cell[0] * .25 + .75)]

    @staticmethod
    def cell_from_pose(pose):
        cell = [-int((pose[1] - 1) / .25), int((pose[0] + 1.5) / .25)]
        cell[0] = 0 if cell[0] < 0 else 7 if cell[0] > 7 else cell[0]
        cell[1] = 0 if cell[1] < 0 else 11 if cell[1] > 11 else cell[1]
        return cell

    def step(self, actions_):
        a = self.args
        self.episode_steps += 1
        info = {}
        message, dist = self.env.step(actions_)
        poses, grid = self.agent_poses, self.grid
        cells = [self.cell_from_pose(poses[:2, i]) for i in range(4)]
        goal = self.pose_from_cell(self.goal_loc)
        obs = []
        for k in range(4):
            self.pixel_type[k] = grid[cells[k][0], cells[k][1]]
            if self.pixel_type[k] == 3:
                self.reached_goal[k] = True
            o = [*poses[:2, k], self.pixel_type[k]]
            for j in self.OTHERS[k]:
                o += [*poses[:2, j], grid[cells[j][0], cells[j][1]]]
            o += goal
            for i in range(2):
                r, c = cells[i]
                left, right = (c - 1 if c > 0 else c), (c + 1 if c < 11 else c)
                up, down = (r - 1 if r > 0 else r), (r + 1 if r < 7 else r)
                o += [grid[up, left], grid[r, left], grid[down, left], grid[up, c], grid[down, c],
                      grid[up, right], grid[r, right], grid[down, right]]
            obs.append(np.array(o, dtype=np.float64))
        if message == '':
            reward = 0
            for k in (2, 3):
                if not self.reached_goal[k]:
                    reward += a.not_reached_penalty
                if self.pixel_type[k] != 3:
                    d = np.linalg.norm(poses[:2, k] - goal)
                    reward += a.dist_multiplier * d ** 2
            terminated = self.episode_steps > a.max_episode_steps
            if not terminated:
                terminated = self.reached_goal[2] and self.reached_goal[3]
        else:
            reward = -30
            terminated = True
            info['message'] = message
        info['dist_travelled'] = dist
        return obs, [reward] * 4, [terminated] * 4, info


PORTS = {"PredatorCapturePrey": PCPPort, "Warehouse": WarehousePort, "MaterialTransport": MTPort,
         "Simple": SimplePort, "ArcticTransport": ArcticPort}


def make_port(scenario, cfg):
    # the restated simulator's collision test is a module-level switch: follow the config,
    # as tests/golden/ref_harness.py does for the reference run
    import rps.robotarium as _rr
    _rr.COLLISION_VARIANT = cfg.get("collision_variant", "offset")
    import rps.utilities.barrier_certificates as _bc   # likewise the stand-in for cvxopt below the certificate closures
    _bc.QP_SOLVER = cfg.get("barrier_solver", "exact")
    return PORTS[scenario](Args(cfg))
