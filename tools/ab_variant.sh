# same-box comparison of the built engine with a variant build: ab_variant.sh <variant.so> [shapes...]
V=$1; shift
for i in 1 2; do
  python tools/size_scaling.py "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/base /"
  NOHUMAN_ENGINE_LIB=$V python tools/size_scaling.py "$@" 2>&1 | grep -v amdgpu.ids | sed "s/^/var  /"
done
