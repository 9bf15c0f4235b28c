"""ORACLE -- restated `rps.utilities.misc` (parity vs real rps unpinned).

Spec: SURVEY.md Appendix A.7.  Reference call sites:
/root/reference/robotarium_gym/utilities/misc.py:7,54 (star import,
generate_initial_conditions), scenarios/Warehouse/warehouse.py:93,
scenarios/*/visualize.py:1 (star import must leak `plt`, `np`,
`determine_marker_size`).
"""
import numpy as np


class _InertPlt(object):
    """Stand-in for matplotlib.pyplot: the visualisers only touch it when a figure is
    shown (never in oracle runs), except `plt.cm.get_cmap` in their constructors."""
    class _CM(object):
        @staticmethod
        def get_cmap(*a, **k):
            return lambda *aa, **kk: (0.0, 0.0, 0.0, 1.0)
    cm = _CM()


plt = _InertPlt()


def generate_initial_conditions(N, spacing=0.3, width=3, height=1.8):
    x_range = int(np.floor(width / spacing))
    y_range = int(np.floor(height / spacing))
    assert x_range != 0, "spacing too large for width"
    assert y_range != 0, "spacing too large for height"
    assert x_range * y_range > N, "too many robots for the grid"
    # upstream shifts the sampled indices by one before divmod: cell (0, 0) is never used, (x_range, 0) is
    choices = (np.random.choice(x_range * y_range, N, replace=False) + 1)
    poses = np.zeros((3, N))
    for i, c in enumerate(choices):
        x, y = divmod(c, y_range)
        poses[0, i] = x * spacing - width / 2
        poses[1, i] = y * spacing - height / 2
        poses[2, i] = np.random.rand() * 2 * np.pi - np.pi
    return poses


def determine_marker_size(robotarium_instance, marker_size_meters):
    return 1.0
