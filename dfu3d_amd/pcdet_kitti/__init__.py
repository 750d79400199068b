"""what OpenPCDet does with the pseudo labels right after they are written
(pcdet/datasets/kitti/kitti_dataset.py: annotations -> LiDAR boxes -> ground-truth database)."""
