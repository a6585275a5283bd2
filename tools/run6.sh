python -m pytest tests/test_gpu_parity.py tests/test_gpu_run.py tests/test_gpu_split.py tests/test_gpu_fullsize.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | head
for i in 1 2; do
  NOHUMAN_ENGINE_LIB=$PWD/tools/old_engine_r02.so python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/OLD  /'
  python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/NEW  /'
  NOHUMAN_SCHED=guided python tools/size_scaling.py pe se 2>&1 | grep -v amdgpu.ids | sed 's/^/NEWgss /'
done
python tools/timeline.py se
NOHUMAN_SCHED=guided python tools/timeline.py se
python tools/timeline.py pe
./tools/bin_bench 193.6 16
./tools/bin_bench 774.4 16
