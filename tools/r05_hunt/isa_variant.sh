#!/bin/bash
# usage: asm_variant.sh <name> <patched device .s>  ->  /root/repo/pyro_amd/libpyrovi_<name>.so (other units: the e1a objects)
set -e
n=$1; s=$2; B=/opt/rocm-7.2.0/lib/llvm/bin
cd /tmp/work/st
$B/clang -cc1as -triple amdgcn-amd-amdhsa -filetype obj -main-file-name lean.hip -target-cpu gfx950 -mrelocation-model pic -o dev_$n.o $s
$B/lld -flavor gnu -m elf64_amdgpu --no-undefined -shared -plugin-opt=-amdgpu-internalize-symbols -plugin-opt=mcpu=gfx950 -plugin-opt=O3 --whole-archive -o dev_$n.out dev_$n.o --no-whole-archive
$B/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=dev_$n.out -output=dev_$n.hipfb
$B/clang-22 -cc1 -triple x86_64-unknown-linux-gnu -aux-triple amdgcn-amd-amdhsa -O3 -emit-llvm-bc -emit-llvm-uselists -disable-free -clear-ast-before-backend -disable-llvm-verifier -discard-value-names -main-file-name lean.hip -mrelocation-model pic -pic-level 2 -fhalf-no-semantic-interposition -mframe-pointer=none -fmath-errno -ffp-contract=off -fno-rounding-math -mconstructor-aliases -funwind-tables=2 -target-cpu x86-64 -tune-cpu generic -resource-dir /opt/rocm-7.2.0/lib/llvm/lib/clang/22 -std=c++17 -fdeprecated-macro -ferror-limit 19 -fhip-new-launch-api -fgnuc-version=4.2.1 -fskip-odr-check-in-gmf -fcxx-exceptions -fexceptions -vectorize-loops -disable-llvm-passes -fcuda-include-gpubinary dev_$n.hipfb -cuid=4d5e85ad268e804f -fcuda-allow-variadic-functions -faddrsig -D__GCC_HAVE_DWARF2_CFI_ASM=1 -o host_$n.bc -x hip-cpp-output lean-host-x86_64-unknown-linux-gnu.hipi
$B/clang++ -O3 -fPIC -c -o lean_$n.o host_$n.bc
hipcc --offload-arch=gfx950 -shared -fPIC -o /root/repo/pyro_amd/libpyrovi_$n.so /root/repo/pyro_amd/_obj/pyrovi.x_e1a.o /root/repo/pyro_amd/_obj/f64.x_e1a.o lean_$n.o
echo built libpyrovi_$n.so
