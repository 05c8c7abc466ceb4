#!/bin/bash
# BVH build variants by environment switch in one call: structural validation + hits, build time at 20 M and 1 M triangles, tree quality
out=gpurun_out/ord; mkdir -p $out
PIMGPU_VALIDATE_BVH=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_config5.py tests/test_gpu_intersect.py tests/test_gpu_scene.py -x -q -m gpu --timeout 250 --timeout-method thread > $out/tests.log 2>&1; echo "tests rc=$?"; tail -3 $out/tests.log
for mode in packed arrays walks; do
  echo "== $mode"
  E="X=1"; [ "$mode" = walks ] && E="PIMGPU_BVH_ORDER_WALKS=1"; [ "$mode" = arrays ] && E="PIMGPU_BVH_PACKED=0"
  env $E timeout -k 10 300 python -u scripts/build_prof.py 20000000 4 2>&1 | tail -1 || exit 1
  env $E timeout -k 10 300 python -u scripts/build_prof.py 1000000 8 2>&1 | tail -1 || exit 1
  env $E PIMGPU_TRACE_DIAG=1 timeout -k 10 200 python -u scripts/raybench.py 1000000 4194304 2>&1 | grep -E "n= 4194304|checksum" || exit 1
done
