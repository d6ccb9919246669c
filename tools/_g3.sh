mkdir -p gpurun_out/g3
timeout 600 python -m pytest tests/test_gpu_traversal.py -q -m gpu > gpurun_out/g3/traversal_checked.log 2>&1
for c in 2 3 4; do timeout 900 python tools/ab_bench.py --config $c --rounds 2 --steps 6 build/libunchecked.so build/libpol0.so build/libpol1.so tracer_amd/lib/libtracer_amd.so build/libpol3.so > gpurun_out/g3/ab_c$c.log 2>&1; done
