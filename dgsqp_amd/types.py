"""Boundary message types of the hot path.

Field-for-field mirror of the dataclasses the reference passes across
``DGSQP.solve()`` (reference DGSQP/types.py:13-37 ``PythonMsg``, :146-166
``Position``/``VehicleActuation``, :202-230 velocity/acceleration/orientation
messages, :325-337 ``ParametricPose``/``ParametricVelocity``, :367-430
``VehicleState``, :463-505 ``VehiclePrediction``).  Only the data layout and the
"no new attributes" guard are part of the boundary; plotting / quaternion
helpers of the reference are not on the path and are not reproduced.
"""
from __future__ import annotations

import array
import copy
from dataclasses import dataclass, field, fields
from typing import Optional


class PythonMsg:
    """Dataclass base whose instances refuse attributes that are not declared fields
    (reference DGSQP/types.py:25-37)."""

    def __setattr__(self, key, value):
        if key not in self.__dataclass_fields__:
            raise TypeError('Cannot add new field "%s" to frozen class %s' % (key, self))
        object.__setattr__(self, key, value)

    def copy(self):
        return copy.deepcopy(self)

    def print(self, depth: int = 0, name: Optional[str] = None):
        pad = '  ' * depth
        head = f'{pad}{name + " (" + type(self).__name__ + ")" if name else type(self).__name__}:\n'
        body = ''
        for f in fields(self):
            v = getattr(self, f.name)
            if isinstance(v, PythonMsg):
                body += v.print(depth + 1, f.name)
            else:
                body += f'{pad}  {f.name}={v}\n'
        if depth == 0:
            print(head + body)
            return None
        return head + body


def _msg(cls):
    return dataclass(cls)


@_msg
class Position(PythonMsg):
    x: float = 0
    y: float = 0
    z: float = 0


@_msg
class VehicleActuation(PythonMsg):
    t: float = 0
    u_a: float = 0
    u_steer: float = 0
    u_ds: float = 0

    def __str__(self):
        return f't:{self.t}, u_a:{self.u_a}, u_steer:{self.u_steer}'


@_msg
class BodyLinearVelocity(PythonMsg):
    v_long: float = 0
    v_tran: float = 0
    v_n: float = 0


@_msg
class BodyAngularVelocity(PythonMsg):
    w_phi: float = 0
    w_theta: float = 0
    w_psi: float = 0


@_msg
class BodyLinearAcceleration(PythonMsg):
    a_long: float = 0
    a_tran: float = 0
    a_n: float = 0


@_msg
class BodyAngularAcceleration(PythonMsg):
    a_phi: float = 0
    a_theta: float = 0
    a_psi: float = 0


@_msg
class OrientationEuler(PythonMsg):
    phi: float = 0
    theta: float = 0
    psi: float = 0


@_msg
class ParametricPose(PythonMsg):
    s: float = 0
    x_tran: float = 0
    n: float = 0
    e_psi: float = 0


@_msg
class ParametricVelocity(PythonMsg):
    ds: float = 0
    dx_tran: float = 0
    dn: float = 0
    de_psi: float = 0


_SUBMSGS = dict(x=Position, v=BodyLinearVelocity, w=BodyAngularVelocity, a=BodyLinearAcceleration,
                aa=BodyAngularAcceleration, e=OrientationEuler, p=ParametricPose, pt=ParametricVelocity,
                u=VehicleActuation)


@_msg
class VehicleState(PythonMsg):
    """Complete vehicle state; sub-messages are created on construction
    (reference DGSQP/types.py:404-421)."""
    t: float = None
    x: Position = None
    v: BodyLinearVelocity = None
    w: BodyAngularVelocity = None
    a: BodyLinearAcceleration = None
    aa: BodyAngularAcceleration = None
    e: OrientationEuler = None
    p: ParametricPose = None
    pt: ParametricVelocity = None
    u: VehicleActuation = None
    du: VehicleActuation = None
    lap_num: int = None

    def __post_init__(self):
        for name, cls in _SUBMSGS.items():
            if getattr(self, name) is None:
                setattr(self, name, cls())


@_msg
class VehiclePrediction(PythonMsg):
    """Predicted trajectory, one ``array.array('d')`` per channel
    (reference DGSQP/types.py:463-505)."""
    t: float = None
    x: array.array = None
    y: array.array = None
    v_x: array.array = None
    v_y: array.array = None
    a_x: array.array = None
    a_y: array.array = None
    psi: array.array = None
    psidot: array.array = None
    v_long: array.array = None
    v_tran: array.array = None
    a_long: array.array = None
    a_tran: array.array = None
    e_psi: array.array = None
    s: array.array = None
    x_tran: array.array = None
    u_a: array.array = None
    u_steer: array.array = None
    u_ds: array.array = None
    lap_num: int = None
