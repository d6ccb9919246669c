#!/bin/bash
# Regenerates the per-round profile set on the GPU box:  gpurun -- 'bash tools/regen_profiles.sh r02'
# writes gpurun_out/<round>/...; copy what is to be kept into profiles/<round>/ (tools/README.md, profiles/r02/README.md).
# The PMC summaries carry the source hash of the library they were collected on; bench.py quotes pmc_config2.json only when
# that hash is the running library's, so the bench line and the as-named lines are taken in a SECOND call, after the summaries
# have been committed:  gpurun -- 'bash tools/regen_profiles.sh r04 lines'
# Round 6: `... r06 pmc` = the PMC set alone; after `lines`, copy gpurun_out/<round>/{bench_line_*.json,configs_as_named.jsonl,pmc_*.json,*_kernel_stats.csv}
# into profiles/<round>/ and run  python3 tools/design_table.py --round <round> --write  -- the ONE table of numbers in DESIGN.md section 8 is generated from them.
R=${1:-r06}; O=gpurun_out/$R; mkdir -p $O
export TMPDIR=/tmp
if [ "$2" = "lines" ]; then      # second call, after the PMC summaries of the first have been copied to profiles/<round>/ and committed
  python3 bench.py --steps 20 --warmup 5 > $O/bench_line_n1.json 2> $O/bench_n1.err          # carries other_configs + hbm_point (round 6)
  python3 bench.py --gpus 2 --steps 2 > $O/bench_line_n2_one_gpu_plumbing.json 2> $O/bench_n2.err
  python3 bench.py --gpus 8 --steps 2 > $O/bench_line_n8_one_gpu_plumbing.json 2> $O/bench_n8.err
  python3 bench.py --gpus 8 --steps 2 --scaling samples --no-other-scaling > $O/bench_line_n8_samples_one_gpu_plumbing.json 2> $O/bench_n8s.err
  for c in 3 4 volume; do python3 tools/config_bench.py --config $c --warmup 8 2> $O/config$c.err; done > $O/configs_as_named.jsonl
  python3 tools/config_bench.py --config 5 2> $O/config5.err >> $O/configs_as_named.jsonl
  exit 0
fi
bash tools/pmc_collect.sh ${R}_c2 python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-fast-math --no-cold --no-other-configs
python3 tools/pmc_summary.py ${R}_c2 "k_render_dense" $O/pmc_config2.json 6 > $O/pmc_config2.txt 2>&1      # 3 timed + 3 vary-seed steps
bash tools/pmc_collect.sh ${R}_c3 python3 tools/config_bench.py --config 3 --spp 32 --steps 3 --warmup 8 --no-cold
python3 tools/pmc_summary.py ${R}_c3 "k_render_pwg<1, false>" $O/pmc_config3.json 3 > $O/pmc_config3.txt 2>&1
bash tools/pmc_collect.sh ${R}_c4 python3 tools/config_bench.py --config 4 --spp 32 --steps 3 --warmup 8 --no-cold
python3 tools/pmc_summary.py ${R}_c4 "k_render_pwg<0, false>" $O/pmc_config4.json 3 > $O/pmc_config4.txt 2>&1
bash tools/pmc_collect.sh ${R}_c5 python3 tools/config_bench.py --config 5 --steps 1 --frames 16
for k in k_sppm_refine k_sppm_camera k_sppm_photon k_sppm_table; do python3 tools/pmc_summary.py ${R}_c5 $k $O/pmc_config5_$k.json > $O/pmc_config5_$k.txt 2>&1; done
bash tools/pmc_collect.sh ${R}_cv python3 tools/config_bench.py --config volume --spp 16 --steps 3 --warmup 8 --no-cold
python3 tools/pmc_summary.py ${R}_cv "k_render_pwg<2, false>" $O/pmc_volume.json 3 > $O/pmc_volume.txt 2>&1
for c in 2 3 4 5 v; do for f in gpurun_out/${R}_c$c/trace/*/*_kernel_stats.csv; do cp $f $O/config${c}_kernel_stats.csv; done; done
bash tools/pmc_collect.sh ${R}_hbm python3 tools/size_sweep.py --k 16 --integrators path --json --steps 4
python3 tools/pmc_summary.py ${R}_hbm "k_render_pwg<0, false>" $O/pmc_hbm_point.json 4 > $O/pmc_hbm_point.txt 2>&1
for f in gpurun_out/${R}_hbm/trace/*/*_kernel_stats.csv; do cp $f $O/hbm_point_kernel_stats.csv; done
if [ "$2" = "pmc" ]; then exit 0; fi      # round 6: the PMC set alone (the summaries are committed, then `lines`)
for c in 3 4 volume; do python3 tools/config_bench.py --config $c --warmup 8 2> $O/config$c.err; done > $O/configs_as_named.jsonl
python3 tools/config_bench.py --config 5 2> $O/config5.err >> $O/configs_as_named.jsonl
python3 bench.py > $O/bench_line_n1.json 2> $O/bench_n1.err
python3 bench.py --gpus 2 --steps 2 > $O/bench_line_n2_one_gpu_plumbing.json 2> $O/bench_n2.err
python3 bench.py --gpus 8 --steps 2 > $O/bench_line_n8_one_gpu_plumbing.json 2> $O/bench_n8.err
python3 bench.py --gpus 8 --steps 2 --scaling samples --no-other-scaling > $O/bench_line_n8_samples_one_gpu_plumbing.json 2> $O/bench_n8s.err
# round 5: throughput across the LDS / L2 / Infinity-Cache / HBM boundaries, the 4 M-triangle operating point under the counters, and
# what FETCH_SIZE / TCC_EA0_RDREQ* count for this path's gathers (tools/probe/gather_probe.hip)
python3 tools/size_sweep.py > $O/size_sweep.txt 2> $O/size_sweep.err
bash tools/fetch_size_calibration.sh $O > $O/fetch_size_calibration.txt 2>&1
python3 tools/tools_profile.py spheres > $O/cycle_profile_config2.txt 2>&1
python3 tools/spp_sweep.py > $O/spp_sweep.txt 2>&1
for c in 2 4 5; do python3 tools/tile_balance.py --config $c > $O/tile_balance_config$c.txt 2> $O/tile_balance_config$c.err; done
python3 tools/tile_balance.py --config 4 --spp 64 > $O/tile_balance_config4_64spp.txt 2>&1
python3 tools/split_trace.py --config 2 --ranks 1,2,4,8 > $O/adaptive_blocks_config2.txt 2>&1
python3 tools/split_trace.py --config 4 --spp 64 --ranks 1,2,8 --launches 6 > $O/adaptive_blocks_config4_64spp.txt 2>&1
python3 tools/lbvh_bench.py > $O/lbvh_bench.txt 2>&1
python3 tools/sah_build_bench.py 6 > $O/sah_build_bench.txt 2>&1
for c in 2 3 4; do python3 tools/cold_start.py --native --config $c --ranks 1,8 --launches 12; done > $O/cold_start.txt 2>&1
for c in 2 3 4; do python3 tools/share_bounds.py --config $c; done > $O/share_bounds.txt 2>&1
python3 tools/split_trace.py --config 3 --ranks 1,8 --launches 14 --vary-seed > $O/sched_trace_config3.txt 2>&1
python3 tools/small_blocks_bench.py > $O/small_blocks.txt 2>&1
python3 tools/sppm_frame_sizes.py > $O/sppm_frame_sizes.txt 2>&1
