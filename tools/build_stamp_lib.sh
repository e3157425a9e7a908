#!/bin/bash
# The product library with in-kernel cycle stamps compiled into k_octree (-DRUMI_OCT_STAMP: one printf per (frame 0, level)) and
# k_orient_desc (-DRUMI_OD_STAMP: every eighth workgroup of frame 0):
#   tools/build_stamp_lib.sh && python tools/with_lib.py tools/bin/librumi_hip_stamp.so tools/oct_stamp.py     (on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); cd $R/rumi_slam_amd/csrc
make -j8 > /dev/null
mkdir -p $R/tools/bin
F="-O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics --offload-arch=gfx950 -I../../include -I."
/opt/rocm/bin/hipcc $F -DRUMI_OCT_STAMP -c orb_octree_kernel.hip -o /tmp/oct_stamp.o
/opt/rocm/bin/hipcc $F -DRUMI_OD_STAMP -c orb_kernels.hip -o /tmp/orbk_stamp.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/librumi_hip_stamp.so /tmp/oct_stamp.o /tmp/orbk_stamp.o $(ls *.o | grep -v -e orb_octree_kernel -e orb_kernels) -lpthread
# k_pose_opt (-DRUMI_POSE_STAMP: one printf per wave of frame 0): ... && python tools/with_lib.py tools/bin/librumi_hip_pose_stamp.so tools/pose_probe.py
/opt/rocm/bin/hipcc $F -DRUMI_POSE_STAMP -c opt.hip -o /tmp/opt_stamp.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/librumi_hip_pose_stamp.so /tmp/opt_stamp.o $(ls *.o | grep -v -e '^opt.o') -lpthread
# k_baw_system (-DRUMI_BAW_STAMP: workgroups 1 and 5 of window 0 print their phases at the third LM trial): ... && python tools/with_lib.py tools/bin/librumi_hip_baw_stamp.so tools/prof_lba_batch.py 0 1
/opt/rocm/bin/hipcc $F -DRUMI_BAW_STAMP -c opt.hip -o /tmp/opt_baw_stamp.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/tools/bin/librumi_hip_baw_stamp.so /tmp/opt_baw_stamp.o $(ls *.o | grep -v -e '^opt.o') -lpthread
