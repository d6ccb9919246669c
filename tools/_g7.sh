mkdir -p gpurun_out/g7
export TMPDIR=/tmp
bash tools/pmc_collect.sh r02_c2 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
python3 tools/pmc_summary.py r02_c2 "k_render<true, false, 0" gpurun_out/g7/pmc_config2.json > gpurun_out/g7/pmc_config2.txt 2>&1
bash tools/pmc_collect.sh r02_c3 python3 tools/config_bench.py --config 3 --spp 32 --steps 3
python3 tools/pmc_summary.py r02_c3 "k_render<false, false, 1" gpurun_out/g7/pmc_config3.json > gpurun_out/g7/pmc_config3.txt 2>&1
bash tools/pmc_collect.sh r02_c4 python3 tools/config_bench.py --config 4 --spp 32 --steps 3
python3 tools/pmc_summary.py r02_c4 "k_render<false, false, 0" gpurun_out/g7/pmc_config4.json > gpurun_out/g7/pmc_config4.txt 2>&1
bash tools/pmc_collect.sh r02_c5 python3 tools/config_bench.py --config 5 --steps 1 --frames 16
for k in k_sppm_refine k_sppm_camera k_sppm_photon k_sppm_hash; do python3 tools/pmc_summary.py r02_c5 $k gpurun_out/g7/pmc_config5_$k.json > gpurun_out/g7/pmc_config5_$k.txt 2>&1; done
for c in 3 4 5 volume; do python3 tools/config_bench.py --config $c > gpurun_out/g7/config$c.json 2>gpurun_out/g7/config$c.err; done
cp gpurun_out/r02_c*/trace/*/*_kernel_stats.csv gpurun_out/g7/ 2>/dev/null
for t in r02_c2 r02_c3 r02_c4 r02_c5; do for f in gpurun_out/$t/trace/*/*_kernel_stats.csv; do cp $f gpurun_out/g7/${t}_kernel_stats.csv; done; done
python3 bench.py --steps 5 --warmup 1 > gpurun_out/g7/bench1.json 2> gpurun_out/g7/bench1.err
python3 tools/tile_balance.py > gpurun_out/g7/tile_balance.txt 2>&1
ls gpurun_out/g7
