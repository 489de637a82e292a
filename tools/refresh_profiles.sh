#!/bin/bash
# Regenerate every artefact under profiles/ for round $1 (e.g. r02) on the GPU box; results land in gpurun_out/profiles_$1/.
#   gpurun --timeout 2400 -- 'bash tools/refresh_profiles.sh r02'      then: cp gpurun_out/profiles_r02/* profiles/
set -u
R=${1:-r02}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
export TMPDIR=/tmp
line() { grep '^{"metric"' | tail -1; }
# 1. the driver's command
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | line > $OUT/${R}_bench_n1.json          # (incl. the `secondary` rows, as the driver runs it)
# 2. rocprofv3 kernel stats of the same command (shorter; no CPU baseline)
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -o r -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
cp $(find gpurun_out/prof_stats -name "*kernel_stats.csv" | head -1) $OUT/${R}_adapt_step_kernel_stats.csv
python3 tools/trace_steps.py gpurun_out/prof_stats 2 --hist > $OUT/${R}_step_trace_graph.json 2>/dev/null
rm -rf gpurun_out/prof_stats
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_eager -o r -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary --graph 0 > /dev/null 2>&1
python3 tools/trace_steps.py gpurun_out/prof_eager > $OUT/${R}_step_trace_eager.json 2>/dev/null
rm -rf gpurun_out/prof_eager
# 3. PMC traffic: separate passes, counters only with --kernel-trace
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --graph 0 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --graph 0 > /dev/null 2>&1
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --steps 3 > $OUT/${R}_pmc_traffic.json 2>/dev/null
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
# 3b. matrix-pipe utilisation per kernel family (its own counter pass)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_mfma -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --graph 0 > /dev/null 2>&1
python3 tools/mfma_busy.py gpurun_out/pmc_mfma --steps 3 > $OUT/${R}_mfma_busy.json 2>/dev/null
rm -rf gpurun_out/pmc_mfma
# 4. the other configurations that fit one GPU, secondary rows
python3 bench.py --mode train --no-cpu-baseline 2>/dev/null | line > $OUT/${R}_bench_n1_configs1_pretrain.json
python3 bench.py --per-gpu-batch 4 --no-cpu-baseline 2>/dev/null | line > $OUT/${R}_bench_n1_configs3_slice.json
python3 bench.py --mode train --encoder 101 --height 1024 --width 2048 --per-gpu-batch 2 --no-cpu-baseline --steps 5 --warmup 2 2>/dev/null | line > $OUT/${R}_bench_n1_configs4_slice.json
python3 bench.py --mode frame --no-cpu-baseline 2>/dev/null | line > $OUT/${R}_bench_n1_frame.json
python3 bench.py --yaml-faithful --no-cpu-baseline --no-roofline --no-secondary 2>/dev/null | line > $OUT/${R}_bench_n1_yaml_faithful.json
python3 tools/peaks.py --json $OUT/${R}_peaks.json > /dev/null 2>&1
# 5. clocks and cycles measured in the kernels (in-kernel stamps; the diagnostic variant of the library: tools/build_variant.sh stamps -DCODEPS_CLOCK_STAMPS)
python3 tools/clock_cycles.py --bare --json $OUT/${R}_clock_cycles_bare.json > /dev/null 2>&1
if [ -f codeps_amd/variants/libcodeps_hip_stamps.so ]; then
  CODEPS_HIP_LIB=$PWD/codeps_amd/variants/libcodeps_hip_stamps.so python3 tools/clock_cycles.py --kernels --json $OUT/${R}_clock_cycles_kernels.json > /dev/null 2>&1
fi
# 6. where the wave cycles of the photometric / weight-gradient / ring / BatchNorm kernels go (one PMC pass)
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/pmc_k -o r -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary --graph 0 > /dev/null 2>&1
python3 tools/pmc_kernels.py gpurun_out/pmc_k photo_ pyramid wgrad_halo64s conv_halo3 conv_igemm chain_ bn_ > $OUT/${R}_pmc_kernels.json 2>/dev/null
rm -rf gpurun_out/pmc_k
# 7. the non-convolution kernels against the bytes they move, the chained 1x1 pair per stage, where the step does not fill the chip
python3 tools/chain_widths.py --json $OUT/${R}_chain_widths.json > /dev/null 2>&1
python3 tools/bn_apply_rate.py --json $OUT/${R}_bn_apply_rate.json > /dev/null 2>&1
python3 tools/dwconv_rate.py --json $OUT/${R}_dwconv_rate.json > /dev/null 2>&1
python3 tools/guda_rate.py --json $OUT/${R}_guda_rate.json > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_cp -o r -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
python3 tools/critical_path.py gpurun_out/trace_cp > $OUT/${R}_critical_path.json 2>/dev/null
rm -rf gpurun_out/trace_cp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c1 -o r -- python3 bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-secondary > /dev/null 2>&1
cp $(find gpurun_out/prof_c1 -name "*kernel_stats.csv" | head -1) $OUT/${R}_configs1_kernel_stats.csv
rm -rf gpurun_out/prof_c1
python3 tools/aten_audit.py --mode train > $OUT/${R}_aten_audit_train.txt 2>/dev/null
python3 tools/aten_audit.py > $OUT/${R}_aten_audit.txt 2>/dev/null
python3 tools/conv_layers.py --json $OUT/${R}_conv_layers.json > $OUT/${R}_conv_layers.txt 2>/dev/null
ls -la $OUT
