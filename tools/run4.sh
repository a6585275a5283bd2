mkdir -p gpurun_out/r03
python -m pytest tests/test_gpu_parity.py tests/test_gpu_run.py tests/test_gpu_split.py -x -q 2>&1 | tail -4
for i in 1 2; do
  NOHUMAN_ENGINE_LIB=$PWD/tools/old_engine_r02.so python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/OLD  /'
  NOHUMAN_SCHED=off python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/NEWoff /'
  python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/NEWgss /'
done
NOHUMAN_SCHED=off python tools/timeline.py se
python tools/timeline.py se
python tools/timeline.py pe
./tools/bin_bench 193.6 18
