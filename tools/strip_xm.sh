#!/bin/bash
# groups-per-XCD sweep of the a-trous tile order on ONE 8K/8 strip (compute only), diag build
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p build
python3 -c "
from svgf_amd import build as b
b.build_library(extra_flags=['-DSVGF_DIAG'], out='$R/build/libsvgf_diag.so')" 2>/dev/null
for round in 1 2; do for xm in default 1 2 4 8 16; do
  echo -n "xm $xm: "
  if [ $xm = default ]; then E=""; else E="SVGF_ATROUS_XM=$xm"; fi
  env SVGF_LIBRARY=$R/build/libsvgf_diag.so $E python3 tools/strip_sim.py --comm none --plan ${PLAN:-ghost} --stream own-hi 2>&1 | grep "ms/frame" | sed 's/.*python: //'
done; done
