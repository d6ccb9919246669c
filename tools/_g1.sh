mkdir -p gpurun_out/g1
nproc > gpurun_out/g1/nproc.txt
# teeth: the unchecked speculative round must FAIL the adversarial traversal test
TRC_AMD_LIB=$PWD/build/libunchecked.so timeout 600 python -m pytest tests/test_gpu_traversal.py -q -m gpu > gpurun_out/g1/traversal_unchecked.log 2>&1
timeout 600 python -m pytest tests/test_gpu_traversal.py -q -m gpu > gpurun_out/g1/traversal_checked.log 2>&1
# A/B of r01 base / unchecked / checked
for c in 2 3 4; do timeout 900 python tools/ab_bench.py --config $c --rounds 2 --steps 6 build/libbase_r01.so build/libunchecked.so tracer_amd/lib/libtracer_amd.so > gpurun_out/g1/ab_c$c.log 2>&1; done
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/g1/bench1.json 2> gpurun_out/g1/bench1.err
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/g1/bench2.json 2> gpurun_out/g1/bench2.err
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/g1/pytest_gpu.log 2>&1
tail -3 gpurun_out/g1/pytest_gpu.log
