"""daccord_amd: MI355X-native (gfx950) implementation of daccord's per-window local de Bruijn
consensus path behind the C ABI in include/daccord_hip.h."""
from ._structs import default_params, DaccParams  # noqa: F401
