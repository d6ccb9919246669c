mkdir -p gpurun_out/g4
for v in pol4 pol5; do TRC_AMD_LIB=$PWD/build/lib$v.so timeout 600 python -m pytest tests/test_gpu_traversal.py -q -m gpu > gpurun_out/g4/traversal_$v.log 2>&1; done
for c in 2 3 4; do timeout 900 python tools/ab_bench.py --config $c --rounds 2 --steps 6 build/libunchecked.so build/libpol0.so build/libpol4.so build/libpol5.so > gpurun_out/g4/ab_c$c.log 2>&1; done
