"""Reference path DGSQP/tracks/track_lib.py (``StraightTrack`` :14, ``CurveTrack`` :27, ``ChicaneTrack`` :54, ``get_track`` :96)
-> dgsqp_amd.tracks.  The scripts use ``from DGSQP.tracks.track_lib import *``."""
from dgsqp_amd.tracks import RadiusArclengthTrack, StraightTrack, CurveTrack, ChicaneTrack, CubicSplineTrack, get_track  # noqa: F401
__all__ = ['RadiusArclengthTrack', 'StraightTrack', 'CurveTrack', 'ChicaneTrack', 'CubicSplineTrack', 'get_track']
