#!/bin/bash
# Round-5 evidence set (run on the GPU box; every profiler pass bounded by `timeout`):
#   1. the bench line, the rocprofv3 --kernel-trace --stats summary of the same command, the RCCL-path line under torchrun (N = 1);
#   2. the headline kernel (resblock_f32w_kernel, B = 512, layer 5): a plain timing pass and PMC passes (SQ + GRBM, FETCH_SIZE,
#      WRITE_SIZE) of the SAME command; the direct-form kernel (f32d) the same way for comparison;
#   3. timing-only ablations and the same-process A/Bs of the F(2,3) block (tools build);
#   4. configs[4]: per conv shape, the A/B of the conv forms, the rocprofv3 --stats split of the whole step, PMC of one w3 layer;
#   5. the white-box gradient step and its kernel split in fp32 and bf16 mode, SQ / traffic counters of the two bf16 backward kernels; 6. the adversarial-operand error table of the fp32-class modes.
#   bash tools/profile_round5.sh [outdir under gpurun_out]        then: python tools/summarize_round5.py <outdir>
set -u
out=${1:-gpurun_out/r5}
repo=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
timeout 1500 python3 "$repo/bench.py" --steps 5 --warmup 1 > "$repo/$out/bench.json" 2> "$repo/$out/bench.err"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/stats" -o r -- python3 "$repo/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-caller-shapes > "$repo/$out/stats.log" 2>&1
cp "$repo/$out"/stats/*kernel_stats.csv "$repo/$out/kernel_stats.csv" 2>/dev/null
timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 "$repo/bench.py" --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs > "$repo/$out/bench_torchrun_n1.json" 2> "$repo/$out/bench_torchrun_n1.err"
bash "$repo/tools/pmc_kernel.sh" "$out/pmc_f32w" resblock_f32w -- "$repo/tools/run_resblock.py" 512 f32 2 > "$repo/$out/pmc_f32w.log" 2>&1
bash "$repo/tools/pmc_kernel.sh" "$out/pmc_f32d" resblock_f32_kernel -- "$repo/tools/run_resblock.py" 512 f32d 2 > "$repo/$out/pmc_f32d.log" 2>&1
( cd "$repo/tools" && timeout 300 python3 ab_f32w.py 256 2>&1 | grep -v amdgpu.ids > "$repo/$out/f32w_ab.txt";
  timeout 300 python3 ab_conv_w3.py 256 2>&1 | grep -v amdgpu.ids > "$repo/$out/conv_w3_ab.txt" )
timeout 600 python3 "$repo/tools/conv_by_shape.py" 256 2>&1 | grep -v amdgpu.ids > "$repo/$out/cfg4_conv_by_shape.txt"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/cfg4_stats" -o r -- python3 "$repo/tools/run_cfg4_step.py" 256 3 > "$repo/$out/cfg4_step.log" 2>&1
cp "$repo/$out"/cfg4_stats/*kernel_stats.csv "$repo/$out/cfg4_kernel_stats.csv" 2>/dev/null
bash "$repo/tools/pmc_kernel.sh" "$out/pmc_w3" conv2d_w3 -- "$repo/tools/run_conv.py" 256 256 16 256 3 4 > "$repo/$out/pmc_w3.log" 2>&1
timeout 300 python3 "$repo/tools/bench_whitebox.py" 10 5 f32 2>&1 | grep -v amdgpu.ids > "$repo/$out/whitebox.txt"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/wb_stats" -o r -- python3 "$repo/tools/bench_whitebox.py" 10 5 f32 > "$repo/$out/wb_stats.log" 2>&1
cp "$repo/$out"/wb_stats/*kernel_stats.csv "$repo/$out/whitebox_kernel_stats.csv" 2>/dev/null
timeout 300 python3 "$repo/tools/bench_whitebox.py" 10 5 bf16 2>&1 | grep -v amdgpu.ids > "$repo/$out/whitebox_bf16.txt"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/wbb_stats" -o r -- python3 "$repo/tools/bench_whitebox.py" 10 5 bf16 > "$repo/$out/wbb_stats.log" 2>&1
cp "$repo/$out"/wbb_stats/*kernel_stats.csv "$repo/$out/whitebox_bf16_kernel_stats.csv" 2>/dev/null
bash "$repo/tools/pmc_kernel.sh" "$out/pmc_bwdb_gate" resblock_bwd_gate_bf16 -- "$repo/tools/time_bwd_bf16.py" 10 > "$repo/$out/pmc_bwdb_gate.log" 2>&1
bash "$repo/tools/pmc_kernel.sh" "$out/pmc_bwdb_conv" resblock_bwd_conv_bf16 -- "$repo/tools/time_bwd_bf16.py" 10 > "$repo/$out/pmc_bwdb_conv.log" 2>&1
timeout 200 python3 "$repo/tools/check_bwd_bf16.py" 2>&1 | grep -v amdgpu.ids > "$repo/$out/bwd_bf16_deviation.txt"
timeout 300 python3 "$repo/tools/adversarial_error.py" f32d f32 f32s 2>&1 | grep -v amdgpu.ids > "$repo/$out/adversarial_error.txt"
ls "$repo/$out"
