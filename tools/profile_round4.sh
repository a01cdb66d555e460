#!/bin/bash
# Round-4 evidence set (run on the GPU box; every profiler pass bounded by `timeout`):
#   1. the bench line, the rocprofv3 --kernel-trace --stats summary of the same command, the RCCL-path line under torchrun (N = 1);
#   2. bf16 mode, deferred-skip form: one eps evaluation at B = 512 (36 block launches + the skip GEMM) -- a plain timing pass and PMC
#      passes (FETCH_SIZE / WRITE_SIZE / SQ counters + GRBM_GUI_ACTIVE) of the SAME command, summed per kernel; the same passes with
#      skip_group 0 (the fused block of round 3) for the like-for-like comparison;
#   3. the go / no-go A/B of the form (tools/ab_bf16_ds.py, B = 512);
#   4. configs[4]: per conv shape, and the rocprofv3 --stats split of the whole step.
#   bash tools/profile_round4.sh [outdir under gpurun_out]        then: python tools/summarize_round4.py <outdir>
set -u
out=${1:-gpurun_out/r4}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
B=512; REPS=2
timeout 1200 python3 "$repo/bench.py" --steps 3 --warmup 1 > "$repo/$out/bench.json" 2> "$repo/$out/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/stats" -o r -- python3 "$repo/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-caller-shapes > "$repo/$out/stats.log" 2>&1
cp "$repo/$out"/stats/*kernel_stats.csv "$repo/$out/kernel_stats.csv" 2>/dev/null
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 "$repo/bench.py" --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs > "$repo/$out/bench_torchrun_n1.json" 2> "$repo/$out/bench_torchrun_n1.err"
for form in ds fused; do
  G=""; [ $form = fused ] && G=0
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$repo/$out/time_$form" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B $REPS $G > "$repo/$out/time_$form.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$repo/$out/fetch_$form" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B $REPS $G > "$repo/$out/fetch_$form.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$repo/$out/write_$form" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B $REPS $G > "$repo/$out/write_$form.log" 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d "$repo/$out/sq_$form" -o r -- python3 "$repo/tools/run_eps_bf16.py" $B $REPS $G > "$repo/$out/sq_$form.log" 2>&1
done
timeout 900 python3 "$repo/tools/ab_bf16_ds.py" 512 2 2>&1 | grep -v amdgpu.ids > "$repo/$out/ab_bf16_ds.txt"
timeout 600 python3 "$repo/tools/conv_by_shape.py" 256 2>&1 | grep -v amdgpu.ids > "$repo/$out/cfg4_conv_by_shape.txt"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/cfg4_stats" -o r -- python3 "$repo/tools/run_cfg4_step.py" 256 3 > "$repo/$out/cfg4_step.log" 2>&1
cp "$repo/$out"/cfg4_stats/*kernel_stats.csv "$repo/$out/cfg4_kernel_stats.csv" 2>/dev/null
#   5. the "board ceiling" of both designs on THIS box: the product kernels (fused and deferred-skip, B = 256) and the synthetic mixes
{ echo "== the product kernels, B = 256 (tools/ab_bf16_ds.py timing part)"; AP_DS_GROUPS=0,36 timeout 600 python3 "$repo/tools/ab_bf16_ds.py" 256 3 | grep -E "round|G = ";
  hipcc --offload-arch=gfx950 -O3 "$repo/tools/micro/mfma_hbm_mix.hip" -o /tmp/mfma_hbm_mix -lpthread 2>/dev/null &&
  { echo "== synthetic mixes, same box, 32000 tiles per launch (a 256-clip launch)"; for b in 1 2; do for m in fused ds_block ds_skip; do timeout 120 /tmp/mfma_hbm_mix 4 $b $m 0; done; done;
    echo "== 16x16x32 in place of 32x32x16"; for m in fused ds_block; do timeout 120 /tmp/mfma_hbm_mix 4 1 $m 1; done; }; } 2>&1 | grep -v amdgpu.ids > "$repo/$out/mfma_hbm_mix.txt"
ls "$repo/$out"
