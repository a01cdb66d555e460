#!/bin/bash
# Round-3 evidence set (run on the GPU box; every profiler pass bounded by `timeout`):
#   1. the bench line, the rocprofv3 --kernel-trace --stats summary of the same command, the RCCL-path line under torchrun (N = 1);
#   2. like-for-like counters: for each arithmetic mode (f32, f32s, f32h, bf16) the residual-block kernel alone at B = 512 on the
#      layers d = 1, 32, 512, 2048 -- one plain timing pass and three PMC passes (FETCH_SIZE / WRITE_SIZE / eight SQ counters +
#      GRBM_GUI_ACTIVE) of the SAME command, split per layer by dispatch order;
#   3. tools-build evidence of the bf16 block: phase stamps, ablations, bit identity of the persistent kernel with the round-1
#      kernel (aligned and ragged lengths), the one-wave-per-SIMD experiment;
#   4. configs[4]: per conv shape, and the rocprofv3 --stats split of the whole step;
#   bash tools/profile_round3.sh [outdir under gpurun_out]        then: python tools/summarize_round3.py <outdir>
set -u
out=${1:-gpurun_out/r3}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
B=512; LAYERS="0 5 9 11"; REPS=3
timeout 1200 python3 "$repo/bench.py" --steps 3 --warmup 1 > "$repo/$out/bench.json" 2> "$repo/$out/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/stats" -o r -- python3 "$repo/bench.py" --steps 1 --warmup 1 --no-cpu-baseline > "$repo/$out/stats.log" 2>&1
cp "$repo/$out"/stats/*kernel_stats.csv "$repo/$out/kernel_stats.csv" 2>/dev/null
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 "$repo/bench.py" --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs > "$repo/$out/bench_torchrun_n1.json" 2> "$repo/$out/bench_torchrun_n1.err"
for prec in f32 f32s f32h bf16; do
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$repo/$out/time_$prec" -o r -- python3 "$repo/tools/run_resblock_layers.py" $B $prec $REPS $LAYERS > "$repo/$out/time_$prec.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$repo/$out/fetch_$prec" -o r -- python3 "$repo/tools/run_resblock_layers.py" $B $prec $REPS $LAYERS > "$repo/$out/fetch_$prec.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$repo/$out/write_$prec" -o r -- python3 "$repo/tools/run_resblock_layers.py" $B $prec $REPS $LAYERS > "$repo/$out/write_$prec.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$repo/$out/sq_$prec" -o r -- python3 "$repo/tools/run_resblock_layers.py" $B $prec $REPS $LAYERS > "$repo/$out/sq_$prec.log" 2>&1
done
timeout 300 python3 "$repo/tools/trace_resblock_bf16p.py" 256 9 > "$repo/$out/phase_trace.txt" 2>&1
timeout 300 python3 "$repo/tools/trace_resblock_bf16p.py" 256 9 0x8000000 > "$repo/$out/kstep_trace.txt" 2>&1
timeout 600 python3 "$repo/tools/dbg_resblock_bf16.py" 256 0 4096 1 2 3 4 8 128 256 384 512 1024 8192 > "$repo/$out/ablation.txt" 2>&1
{ timeout 300 python3 "$repo/tools/cmp_bf16_kernels.py" 4; for L in 130 1001 1002 1003 23457; do echo "L = $L"; AP_CMP_L=$L timeout 300 python3 "$repo/tools/cmp_bf16_kernels.py" 2 0 1 3 5 9 11; done; } > "$repo/$out/cmp_kernels.txt" 2>&1
{ timeout 300 python3 "$repo/tools/ab_bf16w.py" 256 3 2 5 9 11; timeout 300 python3 "$repo/tools/trace_resblock_bf16w.py" 256 9; timeout 300 python3 "$repo/tools/ablate_bf16w.py" 256 9 2; } > "$repo/$out/bf16w_experiment.txt" 2>&1
timeout 600 python3 "$repo/tools/conv_by_shape.py" 256 > "$repo/$out/cfg4_conv_by_shape.txt" 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/cfg4_stats" -o r -- python3 "$repo/tools/run_cfg4_step.py" 256 3 > "$repo/$out/cfg4_step.log" 2>&1
cp "$repo/$out"/cfg4_stats/*kernel_stats.csv "$repo/$out/cfg4_kernel_stats.csv" 2>/dev/null
#   5. power: board power / shader clock per block kernel (random and all-zero activations), the energy table of the bf16 block's
#      ablations, the conv kernels, the operand-image experiment
{ timeout 300 python3 "$repo/tools/power_check.py" 5 256; AP_ZERO=1 timeout 300 python3 "$repo/tools/power_check.py" 4 256; timeout 200 python3 "$repo/tools/power_check_conv.py" 4; } 2>&1 | grep -v amdgpu.ids > "$repo/$out/power_by_mode.txt"
timeout 400 python3 "$repo/tools/power_ablate_bf16.py" 256 3 2>&1 | grep -v amdgpu.ids > "$repo/$out/bf16_energy_ablation.txt"
{ hipcc --offload-arch=gfx950 -O3 "$repo/tools/micro/mfma_power.hip" -o /tmp/mfma_power -lpthread 2>/dev/null && timeout 200 /tmp/mfma_power 5; } > "$repo/$out/mfma_power_calibration.txt" 2>&1
{ hipcc --offload-arch=gfx950 -O3 "$repo/tools/micro/mfma_hbm_mix.hip" -o /tmp/mfma_hbm_mix -lpthread 2>/dev/null && { echo "== one workgroup per CU"; timeout 200 /tmp/mfma_hbm_mix 5 1; echo "== two workgroups per CU"; timeout 200 /tmp/mfma_hbm_mix 5 2; }; } > "$repo/$out/mfma_hbm_mix.txt" 2>&1
timeout 400 python3 "$repo/tools/ab_bf16_ub.py" 256 2 2>&1 | grep -v amdgpu.ids > "$repo/$out/bf16_operand_images_experiment.txt"
ls "$repo/$out"
