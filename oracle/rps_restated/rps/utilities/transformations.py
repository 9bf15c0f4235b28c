"""ORACLE -- restated `rps.utilities.transformations` (parity vs real rps unpinned).
Spec: SURVEY.md Appendix A.5.  Reference call sites: utilities/controller.py:12,21,24."""
import numpy as np


def create_si_to_uni_mapping(projection_distance=0.05, angular_velocity_limit=np.pi):
    def si_to_uni_dyn(dxi, poses):
        M, N = np.shape(dxi)
        cs = np.cos(poses[2, :])
        ss = np.sin(poses[2, :])
        dxu = np.zeros((2, N))
        dxu[0, :] = (cs * dxi[0, :] + ss * dxi[1, :])
        dxu[1, :] = (1 / projection_distance) * (-ss * dxi[0, :] + cs * dxi[1, :])
        dxu[1, dxu[1, :] > angular_velocity_limit] = angular_velocity_limit
        dxu[1, dxu[1, :] < -angular_velocity_limit] = -angular_velocity_limit
        return dxu

    def uni_to_si_states(poses):
        _, N = np.shape(poses)
        si_states = np.zeros((2, N))
        si_states[0, :] = poses[0, :] + projection_distance * np.cos(poses[2, :])
        si_states[1, :] = poses[1, :] + projection_distance * np.sin(poses[2, :])
        return si_states

    return si_to_uni_dyn, uni_to_si_states
