for i in 1 2; do
  NOHUMAN_ENGINE_LIB=$PWD/tools/old_engine_r02.so python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/r02  /'
  python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/r03  /'
done
NOHUMAN_ENGINE_LIB=$PWD/tools/old_engine_r02.so python tools/sweep_sched.py --passes 2 --steps 8 ont hit 2>&1 | grep -v amdgpu.ids | sed 's/^/r02  /'
python tools/sweep_sched.py --passes 2 --steps 8 ont hit 2>&1 | grep -v amdgpu.ids | sed 's/^/r03  /'
