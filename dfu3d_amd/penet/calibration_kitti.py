"""tools/PENet/dataloaders/calibration_kitti.py -> dfu3d_amd.calibration."""
from ..calibration import Calibration, get_calib_from_file  # noqa: F401
