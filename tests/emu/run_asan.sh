#!/bin/bash
# The emulated library under AddressSanitizer (TEST INFRASTRUCTURE; about ten minutes): "device" memory is the sanitizer's heap, so a
# kernel or a host path that touches one element beyond an allocation is reported with its source line.  Every check of
# tests/emu/checks.py and the ABI tour (tools/abi_tour.py: every entry point, error paths included).
#   bash tests/emu/run_asan.sh [check ...]      -> profiles/-style summary on stdout, full log in /tmp/pvi_emu_asan.log
cd "$(dirname "$0")/../.." || exit 1
python tests/emu/build_emu.py --asan > /dev/null || exit 1
B=$PWD/tests/emu/_build
A=$(/opt/rocm/lib/llvm/bin/clang++ -print-file-name=libclang_rt.asan-x86_64.so)
export LD_PRELOAD=$A ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0:halt_on_error=1:detect_stack_use_after_return=0
export PYROVI_LIB=$B/libpyrovi_emu_asan.so PVI_RCCL_LIB=$B/librccl_emu.so PVI_EMU_FULL=1
L=/tmp/pvi_emu_asan.log; : > $L
CHECKS=${*:-"f64_bit_identical f32_paths feedback_4d_and_detector swapped_order feedback_2d_explicit_node slabs_and_halo table_tier_spline_rollout multi_sweep_launches rccl_shards"}
for c in $CHECKS; do
  timeout 2400 python tests/emu/checks.py $c >> $L 2>&1; echo "$c rc=$? :: $(grep -a "== $c passed\|ERROR: AddressSanitizer" $L | tail -1 | cut -c1-160)"
done
timeout 1200 python tools/abi_tour.py >> $L 2>&1; echo "abi_tour rc=$? :: $(grep -a 'ABI-TOUR-OK\|ERROR: AddressSanitizer' $L | tail -1)"
echo "AddressSanitizer reports: $(grep -a -c 'ERROR: AddressSanitizer' $L)"
