"""centroids-reid hot path, MI355X-native (gfx950).  See DESIGN.md."""
from .config import CfgNode, cfg, get_cfg_defaults  # noqa: F401
