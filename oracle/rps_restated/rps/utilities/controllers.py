"""ORACLE -- restated `rps.utilities.controllers` (parity vs real rps unpinned).
Spec: SURVEY.md Appendix A.5.  Reference call site: utilities/controller.py:1,11,22."""
import numpy as np


def create_si_position_controller(x_velocity_gain=1, y_velocity_gain=1, velocity_magnitude_limit=0.15):
    def si_position_controller(xi, positions):
        _, N = np.shape(xi)
        dxi = np.zeros((2, N))
        dxi[0][:] = x_velocity_gain * (positions[0][:] - xi[0][:])
        dxi[1][:] = y_velocity_gain * (positions[1][:] - xi[1][:])
        norms = np.linalg.norm(dxi, axis=0)
        idxs = np.where(norms > velocity_magnitude_limit)
        if norms[idxs].size != 0:
            dxi[:, idxs] *= velocity_magnitude_limit / norms[idxs]
        return dxi
    return si_position_controller
