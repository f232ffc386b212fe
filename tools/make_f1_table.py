"""Build container only: derive the cubic-spline table of the F1 track (BASELINE configs[3]) from the waypoint file of the
reference (/root/reference/DGSQP/tracks/track_data/f1_austin_tenth_scale.npz, loaded by track_lib.get_track :112-113 into
CasadiBSplineTrack).  The reference fits ``ca.interpolant('bspline', degree 3)`` through (s_waypoints, xy_waypoints)
(casadi_bspline_track.py:56-57); CasADi is not available here, the table holds scipy's ``make_interp_spline(k=3)`` interpolant
(not-a-knot ends) as piecewise cubics -- same interpolation conditions, possibly different end conditions (stated deviation,
SURVEY.md Appendix A.2).  Output: dgsqp_amd/track_data/f1_austin_tenth_scale_spline.npz (derived data, not a copy)."""
import pathlib
import numpy as np
from scipy.interpolate import PPoly, make_interp_spline

ROOT = pathlib.Path(__file__).resolve().parent.parent
d = np.load('/root/reference/DGSQP/tracks/track_data/f1_austin_tenth_scale.npz', allow_pickle=True)
assert str(d['save_mode']) == 'casadi_bspline'
s, xy = np.asarray(d['s_waypoints'], float), np.asarray(d['xy_waypoints'], float)
coef = []
for k in range(2):
    pp = PPoly.from_spline(make_interp_spline(s, xy[:, k], k=3))
    # PPoly breakpoints of a not-a-knot spline drop the 2nd and the second-to-last waypoint; re-expand on every waypoint interval
    c = np.zeros((len(s) - 1, 4))
    for i in range(len(s) - 1):
        c[i] = [pp(s[i], nu) / [1, 1, 2, 6][nu] for nu in range(4)]      # ascending powers of (s - s_i)
    coef.append(c)
    assert np.abs(np.polynomial.polynomial.polyval(s[1:] - s[:-1], c.T, tensor=False) - xy[1:, k]).max() < 1e-9
np.savez_compressed(ROOT / 'dgsqp_amd' / 'track_data' / 'f1_austin_tenth_scale_spline.npz', knots=s, cx=coef[0], cy=coef[1],
                    left_width=np.asarray(d['left_width'], float), right_width=np.asarray(d['right_width'], float), slack=2.0)
print('intervals', len(s) - 1, 'length', s[-1] - s[0])
