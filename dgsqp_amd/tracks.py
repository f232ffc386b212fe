"""Track geometry entering the hot path.

Only what ``DGSQP.solve()`` and the Monte-Carlo samplers need from the
reference track classes:

* key points of a radius/arc-length centre line
  (reference DGSQP/tracks/radius_arclength_track.py:361-408),
* the two lookup tables behind ``get_curvature_casadi_fn`` /
  ``get_tangent_angle_casadi_fn`` (:199-225) -- exported as plain arrays that
  the HIP kernels evaluate with CasADi's ``pw_const`` / ``pw_lin`` semantics,
* ``local_to_global`` (:752-807) used by the initial-condition samplers,
* the three parametrised shapes of DGSQP/tracks/track_lib.py:14-87 and the
  ``L_track_barc`` circuit.

Plotting, ``global_to_local`` and the NLP projection are not on the path.
"""
from __future__ import annotations

import numpy as np


def _wrap(theta: float) -> float:
    if theta < -np.pi:
        return 2 * np.pi + theta
    if theta > np.pi:
        return theta - 2 * np.pi
    return theta


class RadiusArclengthTrack:
    """Centre line made of constant-curvature segments ``cl_segs[i] = [length, signed radius]``
    (radius 0 = straight)."""

    def __init__(self, track_width=None, slack=None, cl_segs=None):
        self.track_width = track_width
        self.slack = slack
        self.cl_segs = None if cl_segs is None else np.asarray(cl_segs, dtype=float)
        self.key_pts = None
        self.track_length = None
        self.half_width = None
        self.n_segs = None
        self.circuit = False
        self.phase_out = False

    def initialize(self, track_width=None, slack=None, cl_segs=None, init_pos=(0, 0, 0)):
        if track_width is not None:
            self.track_width = float(track_width)
        if slack is not None:
            self.slack = float(slack)
        if cl_segs is not None:
            self.cl_segs = np.asarray(cl_segs, dtype=float)
        self.half_width = self.track_width / 2
        self.n_segs = self.cl_segs.shape[0]
        self.key_pts = self.get_track_key_pts(self.cl_segs, init_pos)
        self.track_length = float(self.key_pts[-1, 3])
        self.circuit = bool(np.isclose(self.key_pts[0, 0], self.key_pts[-1, 0])
                            and np.isclose(self.key_pts[0, 1], self.key_pts[-1, 1]))
        return self

    @staticmethod
    def get_track_key_pts(cl_segs, init_pos):
        """Rows ``[x, y, psi, cumulative s, segment length, signed curvature]``; row i>0
        describes the END of segment i-1 (radius_arclength_track.py:361-408)."""
        n = cl_segs.shape[0]
        kp = np.zeros((n + 1, 6))
        kp[0, :3] = init_pos
        for i in range(1, n + 1):
            x0, y0, psi0, s0 = kp[i - 1, :4]
            length, r = cl_segs[i - 1]
            if r == 0:
                psi, curv = psi0, 0.0
                x = x0 + length * np.cos(psi0)
                y = y0 + length * np.sin(psi0)
            else:
                xc = x0 - r * np.sin(psi0)
                yc = y0 + r * np.cos(psi0)
                theta = length / r
                x = xc + r * np.sin(psi0 + theta)
                y = yc - r * np.cos(psi0 + theta)
                curv = 1 / r
                psi = _wrap(psi0 + theta)
            kp[i] = [x, y, psi, s0 + length, length, curv]
        return kp

    # ---- tables consumed by the solver (radius_arclength_track.py:199-225) -------------------
    def tables(self):
        """Return ``(L, seg_s[n+1], seg_curv[n], seg_ang[n+1])``:
        curvature(s) = pw_const(sbar, seg_s[1:-1], seg_curv), tangent(s) = pw_lin(sbar, seg_s, seg_ang)."""
        seg_s = self.key_pts[:, 3].copy()
        seg_curv = self.key_pts[1:, 5].copy()
        seg_len = self.key_pts[:, 4]
        curv = self.key_pts[:, 5]
        ang = np.zeros(self.key_pts.shape[0] + 1)
        for i in range(self.key_pts.shape[0]):
            ang[i + 1] = ang[i] if curv[i] == 0 else ang[i] + seg_len[i] * curv[i]
        return self.track_length, seg_s, seg_curv, ang[1:].copy()

    def _sbar(self, s):
        L = self.track_length
        return np.fmod(np.fmod(s, L) + L, L)

    def get_curvature(self, s):
        """Numeric value of the CasADi curvature function (pw_const semantics)."""
        _, seg_s, seg_curv, _ = self.tables()
        sb = self._sbar(s)
        c = seg_curv[0]
        for i in range(len(seg_curv) - 1):
            c = c + (seg_curv[i + 1] - seg_curv[i]) * float(sb >= seg_s[i + 1])
        return c

    def get_tangent_angle(self, s):
        """Numeric value of the CasADi tangent-angle function (pw_lin semantics)."""
        _, seg_s, _, ang = self.tables()
        sb = self._sbar(s)

        def lseg(i):
            return ang[i] + (ang[i + 1] - ang[i]) / (seg_s[i + 1] - seg_s[i]) * (sb - seg_s[i])
        r = lseg(0)
        for i in range(len(seg_s) - 2):
            if sb >= seg_s[i + 1]:
                r = r + (lseg(i + 1) - lseg(i))
        return r

    def get_halfwidth(self, s):
        return self.half_width

    kind = 'arcs'

    def lookup(self, s):
        """Vectorised (curvature, tangent angle) at arclength s."""
        L, seg_s, seg_curv, ang = self.tables()
        sb = np.fmod(np.fmod(s, L) + L, L)
        idx = np.clip(np.searchsorted(seg_s, sb, side='right') - 1, 0, len(seg_curv) - 1)
        slope = (ang[idx + 1] - ang[idx]) / (seg_s[idx + 1] - seg_s[idx])
        return seg_curv[idx], ang[idx] + slope * (sb - seg_s[idx])

    # ---- Frenet -> global (radius_arclength_track.py:752-807) --------------------------------
    def local_to_global(self, cl_coord):
        s, e_y, e_psi = cl_coord
        L = self.track_length
        while s < 0:
            s += L
        while s >= L:
            s -= L
        kp = self.key_pts
        i0 = np.where(s >= kp[:, 3])[0][-1]
        i1 = i0 + 1
        x_s, y_s, psi_s = kp[i0, 0], kp[i0, 1], kp[i0, 2]
        x_f, y_f, psi_f, curv_f = kp[i1, 0], kp[i1, 1], kp[i1, 2], kp[i1, 5]
        seg_len = kp[i1, 4]
        d = s - kp[i0, 3]
        if curv_f == 0:
            x = x_s + (x_f - x_s) * d / seg_len + e_y * np.cos(psi_f + np.pi / 2)
            y = y_s + (y_f - y_s) * d / seg_len + e_y * np.sin(psi_f + np.pi / 2)
            psi = _wrap(psi_f + e_psi)
        else:
            r = 1 / curv_f
            sgn = 1 if r >= 0 else -1
            xc = x_s + abs(r) * np.cos(psi_s + sgn * np.pi / 2)
            yc = y_s + abs(r) * np.sin(psi_s + sgn * np.pi / 2)
            span = d / abs(r)
            psi_d = _wrap(psi_s + sgn * span)
            ang_norm = _wrap(psi_s + sgn * np.pi / 2)
            ang = -(1 if ang_norm >= 0 else -1) * (np.pi - abs(ang_norm))
            x = xc + (abs(r) - sgn * e_y) * np.cos(ang + sgn * span)
            y = yc + (abs(r) - sgn * e_y) * np.sin(ang + sgn * span)
            psi = _wrap(psi_d + e_psi)
        return x, y, psi

    def local_to_global_typed(self, state):
        x, y, psi = self.local_to_global((state.p.s, state.p.x_tran, state.p.e_psi))
        state.x.x, state.x.y, state.e.psi = x, y, psi


class StraightTrack(RadiusArclengthTrack):
    def __init__(self, length, width, slack, phase_out=False):
        segs = [[length, 0], [10, 0]] if phase_out else [[length, 0]]
        super().__init__(width, slack, np.array(segs, dtype=float))
        self.phase_out = phase_out
        self.initialize()
        self.circuit = False


class CurveTrack(RadiusArclengthTrack):
    """straight - arc - straight (track_lib.py:27-52)."""

    def __init__(self, enter_straight_length, curve_length, curve_swept_angle, exit_straight_length,
                 width, slack, phase_out=False, ccw=True):
        sgn = 1 if ccw else -1
        segs = [[enter_straight_length, 0], [curve_length, sgn * curve_length / curve_swept_angle],
                [exit_straight_length, 0]]
        if phase_out:
            segs.append([10, 0])
        super().__init__(width, slack, np.array(segs, dtype=float))
        self.phase_out = phase_out
        self.initialize()
        self.circuit = False


class ChicaneTrack(RadiusArclengthTrack):
    """straight - arc - straight - opposite arc - straight (track_lib.py:54-87)."""

    def __init__(self, enter_straight_length, curve1_length, curve1_swept_angle, mid_straight_length,
                 curve2_length, curve2_swept_angle, exit_straight_length, width, slack,
                 phase_out=False, mirror=False):
        s1, s2 = (1, -1) if mirror else (-1, 1)
        segs = [[enter_straight_length, 0], [curve1_length, s1 * curve1_length / curve1_swept_angle],
                [mid_straight_length, 0], [curve2_length, s2 * curve2_length / curve2_swept_angle],
                [exit_straight_length, 0]]
        if phase_out:
            segs.append([10, 0])
        super().__init__(width, slack, np.array(segs, dtype=float))
        self.phase_out = phase_out
        self.initialize()
        self.circuit = False


# L_track_barc circuit: the numbers of DGSQP/tracks/track_data/L_track_barc.npz
# (cl_segs, track_width, slack) -- data, not code; tests/test_tracks.py checks closure and
# the cumulative lengths quoted in SURVEY.md Appendix B.
_TRACK_DATA = {
    'L_track_barc': dict(
        track_width=1.1, slack=0.3,
        cl_segs=[[2.251, 0.0], [3.620685533262237, 1.1525], [0.9009999999999999, 0.0],
                 [1.6024753617155325, -1.0201675], [0.15000000000000002, 0.0],
                 [3.858449119267546, 1.2281825], [2.25454, 0.0], [1.9173571933848375, 1.2206275],
                 [0.9059049999999997, 0.0]]),
}


class CubicSplineTrack:
    """Centre line given by cubic-spline interpolants x(s), y(s) through arclength-tagged waypoints -- the reference's
    ``CasadiBSplineTrack`` (DGSQP/tracks/casadi_bspline_track.py:10-71) as the solver sees it:
    curvature = (x'y'' - y'x'') / (x'^2 + y'^2)^1.5 (:122-135), tangent angle = atan2(y', x') (:137-149),
    ``sbar = fmod(fmod(s, L) + L, L)``.  The interpolant is held as one cubic per waypoint interval (ascending powers of
    ``s - s_i``); the table is produced by tools/make_f1_table.py with scipy's ``make_interp_spline(k=3)`` -- CasADi's
    ``interpolant('bspline')`` satisfies the same interpolation conditions, its end conditions may differ (stated deviation)."""
    kind = 'spline'

    def __init__(self, knots, cx, cy, left_width, right_width, slack):
        self.knots = np.asarray(knots, float)
        self.cx, self.cy = np.asarray(cx, float), np.asarray(cy, float)
        self.left_width_points, self.right_width_points = np.asarray(left_width, float), np.asarray(right_width, float)
        self.track_width = float(np.mean(self.left_width_points + self.right_width_points))      # casadi_bspline_track.py:20-21
        self.half_width = self.track_width / 2
        self.slack = slack
        self.track_length = float(self.knots[-1] - self.knots[0])
        self.circuit = bool(np.allclose([self.cx[0, 0], self.cy[0, 0]], self._xy(self.knots[-1] - 1e-12), atol=1e-6))

    def _sbar(self, s):
        L = self.track_length
        return np.fmod(np.fmod(s, L) + L, L)

    def _piece(self, sb):
        i = np.clip(np.searchsorted(self.knots, sb, side='right') - 1, 0, len(self.knots) - 2)
        return i, sb - self.knots[i]

    def _xy(self, s):
        i, t = self._piece(self._sbar(np.asarray(s, float)))
        return (((self.cx[i, 3] * t + self.cx[i, 2]) * t + self.cx[i, 1]) * t + self.cx[i, 0],
                ((self.cy[i, 3] * t + self.cy[i, 2]) * t + self.cy[i, 1]) * t + self.cy[i, 0])

    def _derivs(self, s):
        i, t = self._piece(self._sbar(np.asarray(s, float)))
        dx = (3 * self.cx[i, 3] * t + 2 * self.cx[i, 2]) * t + self.cx[i, 1]
        dy = (3 * self.cy[i, 3] * t + 2 * self.cy[i, 2]) * t + self.cy[i, 1]
        return dx, dy, 6 * self.cx[i, 3] * t + 2 * self.cx[i, 2], 6 * self.cy[i, 3] * t + 2 * self.cy[i, 2]

    def lookup(self, s):
        """Vectorised (curvature, tangent angle) at arclength s."""
        dx, dy, ddx, ddy = self._derivs(s)
        return (dx * ddy - dy * ddx) / np.power(dx ** 2 + dy ** 2, 1.5), np.arctan2(dy, dx)

    def get_curvature(self, s):
        return float(self.lookup(s)[0])

    def get_tangent_angle(self, s):
        return float(self.lookup(s)[1])

    def get_halfwidth(self, s):
        return self.half_width

    def local_to_global(self, cl_coord):
        """casadi_bspline_track.py:151-170: centre-line point plus e_y along the left normal, heading = tangent + e_psi."""
        s, e_y, e_psi = cl_coord
        x, y = self._xy(s)
        dx, dy, _, _ = self._derivs(s)
        nrm = np.hypot(dx, dy)
        return float(x - e_y * dy / nrm), float(y + e_y * dx / nrm), float(np.arctan2(dy, dx) + e_psi)

    def local_to_global_typed(self, state):
        x, y, psi = self.local_to_global((state.p.s, state.p.x_tran, state.p.e_psi))
        state.x.x, state.x.y, state.e.psi = x, y, psi

    def spline_table(self):
        """[knots (n), x coefficients (n-1) x 4, y coefficients (n-1) x 4] as the C-ABI takes it (dgsqp_problem_t.spline)."""
        return np.ascontiguousarray(np.concatenate([self.knots - self.knots[0], self.cx.ravel(), self.cy.ravel()]))


_SPLINE_TRACKS = {'f1_austin_tenth_scale': 'f1_austin_tenth_scale_spline.npz'}


def get_track(name: str):
    key = name[:-4] if name.endswith('.npz') else name
    if key in _SPLINE_TRACKS:
        import pathlib
        d = np.load(pathlib.Path(__file__).resolve().parent / 'track_data' / _SPLINE_TRACKS[key])
        return CubicSplineTrack(d['knots'], d['cx'], d['cy'], d['left_width'], d['right_width'], float(d['slack']))
    if key not in _TRACK_DATA:
        raise ValueError('Chosen Track is unavailable: %s\n Available Tracks: %s' % (name, sorted(_TRACK_DATA)))
    d = _TRACK_DATA[key]
    return RadiusArclengthTrack().initialize(d['track_width'], d['slack'], np.array(d['cl_segs'], dtype=float))
