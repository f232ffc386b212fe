"""Reference path DGSQP/tracks/radius_arclength_track.py -> dgsqp_amd.tracks.RadiusArclengthTrack."""
from dgsqp_amd.tracks import RadiusArclengthTrack  # noqa: F401
