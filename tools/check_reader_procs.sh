#!/bin/bash
# GPU box: the CLI with forked reader processes writes byte-identical label / virtual-point files to the thread path.
set -e
D=/dev/shm/dfu3d_rp
rm -rf $D
python3 - <<'PY'
import numpy as np
from dfu3d_amd import kitti_io, synth
from dfu3d_amd.params import NUSC_CLASSES
H, W, M = 180, 320, 5
for f in range(7):
    s = synth.make_scene(400 + f, H=H, W=W, M=M, cams=1, dense=True, k_min=12, k_max=16)
    n = int(s.n_inst[0])
    kitti_io.write_frame("/dev/shm/dfu3d_rp", f, s.points.numpy(), s.calibs[0], np.full((H, W, 3), 9 * f, np.uint8),
                         s.masks[0][:n].numpy(), s.inst_class[0][:n].numpy(), s.inst_score[0][:n].numpy(),
                         s.inst_box[0][:n].numpy(), NUSC_CLASSES, s.depth[0].numpy(), compress=bool(f % 2))
PY
python3 -m dfu3d_amd.penet.main --detpath $D --batch-frames 4 --reader-procs 3 --conf_files x.yaml
mv $D/label_2 $D/label_procs; mv $D/velodyne_depth $D/vd_procs
python3 -m dfu3d_amd.penet.main --detpath $D --batch-frames 4 --reader-procs 0 --conf_files x.yaml
for f in $D/label_2/*.txt; do cmp $f $D/label_procs/$(basename $f); done
for f in $D/velodyne_depth/*.npy; do cmp $f $D/vd_procs/$(basename $f); done
echo "reader processes == reader threads: $(ls $D/label_2 | wc -l) label files, $(cat $D/label_2/*.txt | wc -l) boxes"
rm -rf $D
