#!/bin/bash
# tools/isa.sh <kernel-name-substring> : device ISA + resource usage of one kernel of csrc/arah_hip.hip -> /tmp/<name>.s
cd /root/repo/arah_release_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -Xclang -target-feature -Xclang -packed-fp32-ops ${ARAH_SLP:--fno-slp-vectorize} $ISA_FLAGS -S --cuda-device-only arah_hip.hip -o /tmp/arah.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -v "not a recognized" | grep -A10 "Function Name: .*$1" | grep -v "^ *[0-9]* |\|^ *| " | head -${2:-14}
awk "/^_ZN[^:]*$1[^:]*:/,/s_endpgm/" /tmp/arah.s > /tmp/$1.s
echo "lines $(wc -l < /tmp/$1.s) mfma $(grep -c v_mfma /tmp/$1.s) scratch $(grep -c scratch_ /tmp/$1.s) branches $(grep -c s_cbranch /tmp/$1.s) waitcnt $(grep -c s_waitcnt /tmp/$1.s)"
