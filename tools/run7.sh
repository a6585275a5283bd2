for i in 1 2; do
  NOHUMAN_ENGINE_LIB=$PWD/tools/old_engine_r02.so python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/OLD    /'
  python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/NEW    /'
  NOHUMAN_ENGINE_LIB=$PWD/tools/var_w1.so python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/W1     /'
  NOHUMAN_ENGINE_LIB=$PWD/tools/var_cdir.so python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/CDIR   /'
  NOHUMAN_ENGINE_LIB=$PWD/tools/var_w1cdir.so python tools/size_scaling.py pe 2>&1 | grep -v amdgpu.ids | sed 's/^/W1CDIR /'
done
