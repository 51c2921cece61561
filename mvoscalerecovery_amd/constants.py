"""Named constants of the hot path (all literals of /root/reference/src/scale_calculator.py)
and the per-frame status codes shared with include/mvosr.h."""
import numpy as np

CAMERA_PITCH = -0.5 * np.pi / 180      # scale_calculator.py:24
VANISH = 185                           # :22
FOCUS = 718                            # :22
PITCH_THRESHOLD_DEG = -80.0            # :235,:239
SKEW_THRESHOLD = 0.3                   # :348
MODE_REL = 0.33                        # :461
MODE_MIN = 2                           # :451,:462
HIST_BINS = 169                        # :326 (170 edges k*0.1)

# enum mvosr_status
ST_MODE = 0
ST_RIGHT = 1
ST_MEDIAN = 2
ST_LEVEL = 3
ST_NO_FLAT = 4
ST_ERR_LEFT = 5
ST_ERR_RIGHT = 6
ST_ERR_SINGULAR = 7
ST_ERR_MASK = 8
ST_ERR_EMPTY = 9
ST_TOO_FEW = 10
ST_RS_FEW = 11                         # rescale variant: fewer than 12 selected points, the previous scale is pushed (rescale.py:152,175)

STATUS_NAMES = {ST_MODE: "mode", ST_RIGHT: "right-edge (skew)", ST_MEDIAN: "median (no modes)",
                ST_LEVEL: "height_level (no points left)", ST_NO_FLAT: "no flat feature",
                ST_ERR_LEFT: "IndexError (left)", ST_ERR_RIGHT: "IndexError (right)",
                ST_ERR_SINGULAR: "LinAlgError (singular triangle)", ST_ERR_MASK: "inconsistent triangulation",
                ST_ERR_EMPTY: "empty frame", ST_TOO_FEW: "too few features below the vanishing row (previous height_level)"}
ERROR_STATUSES = (ST_ERR_LEFT, ST_ERR_RIGHT, ST_ERR_SINGULAR, ST_ERR_MASK, ST_ERR_EMPTY)   # the frame raises

# enum mvosr_count_slot
CNT_VALID, CNT_TRI_PITCH, CNT_TRI_VALID, CNT_SELECTED, CNT_KEPT, CNT_MODES, CNT_MODE_LEFT, CNT_MODE_RIGHT = range(8)
