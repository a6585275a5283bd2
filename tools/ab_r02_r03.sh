# same-box comparison of the round-2 kernels (tools/old_engine_r02.so, built from commit 0a1d40a) with the current ones
for i in 1 2; do
  NOHUMAN_ENGINE_LIB=$PWD/tools/old_engine_r02.so python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/r02  /'
  python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/r03  /'
  NOHUMAN_SCHED=off python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/r03flat /'
done
