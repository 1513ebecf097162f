"""visual_slam_amd -- MI355X-native visual-odometry front-end (drop-in for the hot path of
gilbertgonz/visual_slam: libs/vo_utils.py).  See DESIGN.md and INTEGRATION.md."""
from .vo_utils import VisualOdometry  # noqa: F401
