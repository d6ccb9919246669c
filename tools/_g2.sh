mkdir -p gpurun_out/g2
timeout 600 python -m pytest tests/test_gpu_traversal.py -q -m gpu > gpurun_out/g2/traversal_checked.log 2>&1
for c in 2 3 4; do timeout 900 python tools/ab_bench.py --config $c --rounds 2 --steps 6 build/libunchecked.so tracer_amd/lib/libtracer_amd.so build/libapproxdiv.so > gpurun_out/g2/ab_c$c.log 2>&1; done
timeout 3000 python -m pytest tests -m gpu -x -q > gpurun_out/g2/pytest_gpu.log 2>&1
tail -3 gpurun_out/g2/pytest_gpu.log
