#!/bin/bash
# tools/dev/mkvariant.sh <tag> [extra hipcc flags...] — developer aid: lib/librplgpu_<tag>.so with
# rpl_voxel.hip compiled under the given -D flags (the other translation units are compiled once
# into build/obj and reused), for same-box A/B timing runs (tools/dev/ab.sh).
set -e
R=$(cd $(dirname $0)/../.. && pwd); C=$R/rplidar_ros2_driver_amd/csrc; O=$R/build/obj; mkdir -p $O
TAG=$1; shift
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -I$R/include -I$C"
for f in rpl_kernels rpl_laserscan rpl_ror rpl_decode rpl_msg rpl_fuse rpl_project rplgpu_api; do
  if [ ! -f $O/$f.o ] || [ $C/$f.hip -nt $O/$f.o ] || [ $C/rpl_device.hpp -nt $O/$f.o ] || [ $C/rpl_launch.hpp -nt $O/$f.o ]; then
    /opt/rocm/bin/hipcc $FL -c $C/$f.hip -o $O/$f.o &
  fi
done
/opt/rocm/bin/hipcc $FL "$@" -c $C/rpl_voxel.hip -o $O/rpl_voxel_$TAG.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $R/rplidar_ros2_driver_amd/lib/librplgpu_$TAG.so $O/rpl_kernels.o $O/rpl_laserscan.o $O/rpl_ror.o $O/rpl_decode.o $O/rpl_msg.o $O/rpl_fuse.o $O/rpl_project.o $O/rplgpu_api.o $O/rpl_voxel_$TAG.o
echo built librplgpu_$TAG.so
