#!/bin/bash
# WRITE_SIZE against known byte counts in the store patterns of the 4-D window sweep (tools/writecal.hip) -> gpurun_out/r06_writecal.log
cd /root/repo; mkdir -p gpurun_out; O=gpurun_out/r06_writecal.log; : > $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/writecal tools/writecal.hip || exit 1
for shape in "151 19 26 8000" "101 10 51 20000" "151 19 32 8000"; do
  rm -rf /tmp/wc; echo "== $shape" >> $O
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/wc -o p -- /tmp/writecal $shape 3 > /tmp/wc.log 2>&1)
  grep -E "^V |^WROTE" /tmp/wc.log >> $O
  python tools/writecal_summary.py /tmp/wc /tmp/wc.log >> $O 2>&1
done
cat $O
