"""Canonical parameters of the pseudo-box path (nuScenes variant of the
reference, tools/PENet/dataloaders/my_loader.py).
All literals that the reference hard-codes are exposed here with the
reference's values as defaults."""
from dataclasses import dataclass
from typing import Tuple

import numpy as np

NUSC_CLASSES = ['Car', 'Truck', 'Construction_vehicle', 'Bus', 'Trailer', 'Barrier',
                'Motorcycle', 'Bicycle', 'Pedestrian', 'Traffic_cone']     # my_loader.py:634
VEHICLE_CLASSES = ('Car', 'Truck', 'Construction_vehicle', 'Bus', 'Trailer')     # :581
SMALL_CLASSES = ('Pedestrian', 'Bicycle', 'Motorcycle', 'Barrier', 'Traffic_cone')  # :591


@dataclass
class Params:
    depth_min: float = 0.001            # my_loader.py:507
    plane_max_hs: float = 8.0           # :448
    plane_range: float = 54.0           # :448
    plane_offset: float = 0.05          # :471
    bounds_hw: Tuple[int, int] = (900, 1600)   # :526 (hard-coded pixel bounds)
    fov_hw: Tuple[int, int] = (900, 1600)      # vis_utils.py:153 (hazard H11)
    z_max: float = 1.0                  # :540
    theta_min: float = 1.5              # :175
    # spconv keeps voxel size / range as float32 and promotes to double
    vsize: Tuple[float, float, float] = (float(np.float32(200.0)), float(np.float32(0.002)),
                                         float(np.float32(0.002)))        # :70
    vrange_min: Tuple[float, float, float] = (-100.0, -5.0, -5.0)          # :71
    vgrid: Tuple[int, int, int] = (1, 5000, 5000)
    max_points_per_voxel: int = 100     # :73
    max_voxels: int = 1000000           # :74
    nb_points: int = 1                  # :584
    radius_vehicle: float = 3.0         # :584
    radius_small: float = 0.6           # :593
    small_class_drop_lidar: bool = True  # hazard H4 (:592-594)
    fuse_C: float = 0.1                 # :603
    R0: float = 3.0                     # rectangle_fitting.py:36
    Rd: float = 0.001                   # rectangle_fitting.py:37
    dtheta_deg: float = 1.0             # rectangle_fitting.py:35
    car_aspect_max: float = 5.0         # :651
    stat_filter: bool = False           # dormant: my_loader0.py:734-735
    stat_voxel: float = 0.05            # voxel_down_sample in front of it (my_loader0.py:734)
    stat_nb_neighbors: int = 30
    stat_std_ratio: float = 0.3
    ransac_trials: int = 100            # sklearn RANSACRegressor default max_trials
    ransac_seed: int = 0                # hazard H1

    def thetas(self):
        """Candidate headings of rectangle_fitting.py:119 -> (n_theta, dtheta)."""
        dtheta = float(np.deg2rad(self.dtheta_deg))
        n = len(np.arange(0.0, np.pi / 2.0 - dtheta, dtheta))
        return n, dtheta

    def instance_radii(self, seem_name):
        """(radius for LiDAR points, radius for pseudo points): >0 filter,
        0 keep all (hazard H5), <0 drop all (hazard H4).  my_loader.py:581-599."""
        if seem_name in VEHICLE_CLASSES:
            return self.radius_vehicle, self.radius_vehicle
        if seem_name in SMALL_CLASSES:
            return (-1.0 if self.small_class_drop_lidar else self.radius_small), self.radius_small
        return 0.0, 0.0
