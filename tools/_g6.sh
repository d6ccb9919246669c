mkdir -p gpurun_out/g6
for c in 2 3 4; do timeout 900 python tools/ab_bench.py --config $c --rounds 2 --steps 6 build/libunchecked.so tracer_amd/lib/libtracer_amd.so build/libdmL2G2.so build/libdmL4G4.so build/libdmL3G6.so build/libdmL1G12.so > gpurun_out/g6/ab_c$c.log 2>&1; done
