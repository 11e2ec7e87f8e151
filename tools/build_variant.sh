#!/bin/bash
# A/B builds of libqenv.so: tools/build_variant.sh NAME "<extra hipcc flags>" [unit.hip ...]
# Recompiles the named translation units (default: qenv_ppo.hip) with the extra flags and links them with the product's
# other objects into build_variants/libqenv_NAME.so (git-ignored, travels to the GPU box); select it with QENV_LIB=.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; FLAGS=$2; shift 2 || true
UNITS=${@:-qenv_ppo.hip}
CSRC=$ROOT/autonomous_quadrotor_environment_amd/csrc
OUT=$ROOT/build_variants; mkdir -p $OUT/obj_$NAME
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import __graft_entry__ as g; g.build()" >/dev/null
OBJS=""
for u in qenv_env qenv_env_mix qenv_env_f64 qenv_env_lean qenv_ppo qenv_train; do
  if echo " $UNITS " | grep -q " $u.hip "; then
    UF=""; case $u in qenv_env*) UF="-mllvm -amdgpu-mfma-vgpr-form";; esac   # (as __graft_entry__.UNIT_FLAGS)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -ffp-contract=on -Wno-unused-function $UF $FLAGS -c $CSRC/$u.hip -o $OUT/obj_$NAME/$u.o &
    OBJS="$OBJS $OUT/obj_$NAME/$u.o"
  else
    OBJS="$OBJS $CSRC/_build/$u.o"
  fi
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libqenv_$NAME.so $OBJS
echo $OUT/libqenv_$NAME.so
