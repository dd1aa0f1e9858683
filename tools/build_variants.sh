#!/bin/bash
# tools/build_variants.sh "<name>:<-D flags>" ...  -- A/B builds of libezkl_hip.so with compile-time switches, as ab/libezkl_hip_<name>.so
# (git-ignored, shipped to the GPU box; selected with EZKL_HIP_LIB, see tools/msm_ab.py).  Only the objects named in OBJS (default: msm) are
# rebuilt with the flags, the rest come from the default build.
#   bash tools/build_variants.sh "base:-DEZKL_MSM_PREFETCH=0 -DEZKL_FUSED_Y=0" "pf:-DEZKL_FUSED_Y=0"
R=$(cd "$(dirname "$0")/.." && pwd); C="$R/ezkl_amd/csrc"; mkdir -p "$R/ab"
make -C "$C" -j8 >/dev/null || exit 1
OBJS=${OBJS:-msm}
for V in "$@"; do
  N=${V%%:*}; F=${V#*:}; D="$R/ab/obj_$N"; mkdir -p "$D"
  LINK=""
  for O in capi ntt msm vecops evalh ubench comm g2; do
    if [[ " $OBJS " == *" $O "* ]]; then
      /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-unused-result -Wno-psabi $F -c "$C/$O.hip" -o "$D/$O.o" || exit 1
      LINK="$LINK $D/$O.o"
    else LINK="$LINK $C/$O.o"; fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$R/ab/libezkl_hip_$N.so" $LINK -lhiprtc -L/opt/rocm/lib -lrocprofiler-sdk-roctx -ldl || exit 1
  rm -rf "$D"; echo "built ab/libezkl_hip_$N.so ($F)"
done
