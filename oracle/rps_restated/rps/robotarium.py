"""ORACLE -- restated `rps.robotarium.Robotarium` (parity vs real rps unpinned).

Spec: SURVEY.md Appendix A.1-A.4.  Call sites in the reference:
/root/reference/robotarium_gym/utilities/roboEnv.py:54 (get_poses), :65
(set_velocities), :78 (step), :84-91 (_errors), :109-112 (ctor + first step),
:121 (call_at_scripts_end).
"""
import numpy as np

# Collision test variant:
#   'offset' : centres shifted collision_offset along the heading, distance <= collision_diameter
#   'center' : plain centre distance <= robot_diameter
# The reference reads `_errors['collision'].values()` (roboEnv.py:84), i.e. per-robot
# dict counters; upstream introduced those together with the offset test, so
# 'offset' is the default of sim_spec_v0.  Selectable for tests.
COLLISION_VARIANT = 'offset'

# upstream keeps the counters in a mutable default argument of _validate, so they
# accumulate for the life of the process (Appendix A.4).  Module global here.
_ERRORS = {}


class Robotarium(object):
    def __init__(self, number_of_robots=-1, show_figure=True, sim_in_real_time=True,
                 initial_conditions=np.array([])):
        self.number_of_robots = number_of_robots
        self.show_figure = show_figure
        self.initial_conditions = initial_conditions
        self.boundaries = [-1.6, -1, 3.2, 2]
        self.time_step = 0.033
        self.robot_diameter = 0.11
        self.wheel_radius = 0.016
        self.base_length = 0.105
        self.max_linear_velocity = 0.2
        self.max_angular_velocity = 2 * (self.wheel_radius / self.robot_diameter) * \
            (self.max_linear_velocity / self.wheel_radius)
        self.max_wheel_velocity = self.max_linear_velocity / self.wheel_radius
        self.collision_offset = 0.025
        self.collision_diameter = 0.135
        self.velocities = np.zeros((2, number_of_robots))
        self.poses = self.initial_conditions      # aliases the caller's array (A.1)
        self.figure = None
        self.axes = None
        self._called_step_already = True
        self._checked_poses_already = False
        self._errors = {}
        self._iterations = 0

    # -- A.3
    def set_velocities(self, ids, velocities):
        idxs = np.where(np.abs(velocities[0, :]) > self.max_linear_velocity)
        velocities[0, idxs] = self.max_linear_velocity * np.sign(velocities[0, idxs])
        idxs = np.where(np.abs(velocities[1, :]) > self.max_angular_velocity)
        velocities[1, idxs] = self.max_angular_velocity * np.sign(velocities[1, idxs])
        self.velocities = velocities

    # -- A.2: returns the live array, no copy
    def get_poses(self):
        assert not self._checked_poses_already, "Can only call get_poses() once per call of step()."
        self._called_step_already = False
        self._checked_poses_already = True
        return self.poses

    def call_at_scripts_end(self):
        pass

    # -- A.4
    def step(self):
        assert not self._called_step_already, "Make sure to call get_poses before calling step() again."
        self._called_step_already = True
        self._checked_poses_already = False
        self._errors = self._validate()
        self._iterations += 1
        p = self.poses
        v = self.velocities
        p[0, :] = p[0, :] + self.time_step * np.cos(p[2, :]) * v[0, :]
        p[1, :] = p[1, :] + self.time_step * np.sin(p[2, :]) * v[0, :]
        p[2, :] = p[2, :] + self.time_step * v[1, :]
        p[2, :] = np.arctan2(np.sin(p[2, :]), np.cos(p[2, :]))

    def _validate(self, errors=None):
        if errors is None:
            errors = _ERRORS
        p = self.poses
        b = self.boundaries
        N = self.number_of_robots
        for i in range(N):
            x = p[0, i]
            y = p[1, i]
            if x < b[0] or x > (b[0] + b[2]) or y < b[1] or y > (b[1] + b[3]):
                d = errors.setdefault("boundary", {})
                d[i] = d.get(i, 0) + 1
        for j in range(N - 1):
            for k in range(j + 1, N):
                if COLLISION_VARIANT == 'offset':
                    first = p[:2, j] + self.collision_offset * np.array([np.cos(p[2, j]), np.sin(p[2, j])])
                    second = p[:2, k] + self.collision_offset * np.array([np.cos(p[2, k]), np.sin(p[2, k])])
                    hit = np.linalg.norm(first - second) <= self.collision_diameter
                else:
                    hit = np.linalg.norm(p[:2, j] - p[:2, k]) <= self.robot_diameter
                if hit:
                    d = errors.setdefault("collision", {})
                    d[j] = d.get(j, 0) + 1
                    d[k] = d.get(k, 0) + 1
        return errors
