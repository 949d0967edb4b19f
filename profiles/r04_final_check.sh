mkdir -p gpurun_out/r4l
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4l/smoke.log 2>&1; tail -1 gpurun_out/r4l/smoke.log
python -m pytest tests -m gpu -x -q > gpurun_out/r4l/gpu_tests.log 2>&1; tail -3 gpurun_out/r4l/gpu_tests.log
python3 bench.py > gpurun_out/r4l/bench_default.json 2> gpurun_out/r4l/bench_default.log; tail -c 600 gpurun_out/r4l/bench_default.json
