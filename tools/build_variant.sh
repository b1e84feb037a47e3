#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: the product source with extra -D flags -> tools/ubench/bin/libarah_<name>.so
NAME=$1; shift
mkdir -p /root/repo/tools/ubench/bin
cd /root/repo/arah_release_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Xclang -target-feature -Xclang -packed-fp32-ops ${ARAH_SLP:--fno-slp-vectorize} "$@" -shared -fPIC arah_hip.hip -o /root/repo/tools/ubench/bin/libarah_$NAME.so 2>&1 | grep -v "not a recognized"
ls -la /root/repo/tools/ubench/bin/libarah_$NAME.so
