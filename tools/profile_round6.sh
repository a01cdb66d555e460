#!/bin/bash
# Round-6 evidence set (run on the GPU box; every profiler pass bounded by `timeout`; --pmc passes are their own runs, never combined
# with other trace domains):
#   1. the bench line, the rocprofv3 --kernel-trace --stats summary of the same command, the RCCL-path line under torchrun (N = 1);
#   2. the headline kernel (resblock_f32w_kernel, B = 512, layer 5): timing + PMC passes (SQ + GRBM, FETCH_SIZE, WRITE_SIZE) -- re-measured,
#      not quoted from round 5;
#   3. both bf16 modes (AP_PREC_BF16, AP_PREC_BF16_STORE): per-kernel times and traffic of one eps evaluation at B = 512 (tools/profile_bf16_modes.sh);
#   4. the AP_PREC_F32_SPLIT forms (direct / F(2,3) two-kernel): SQ counters + same-process timing;
#   5. the white-box gradient step in fp32 and bf16 (bf16: kept gate factors) with its kernel split; SQ / traffic of the two bf16 backward kernels;
#   6. configs[4]: per conv shape + the rocprofv3 --stats split of the whole step.
#   bash tools/profile_round6.sh [outdir under gpurun_out]        then: python tools/summarize_round6.py <outdir>
set -u
out=${1:-gpurun_out/r6}
repo=$(cd "$(dirname "$0")/.." && pwd)
legs=${LEGS:-bench f32w bf16 f32s wb cfg4 adv}           # LEGS="bench cfg4" re-runs a subset into the same directory
want() { case " $legs " in *" $1 "*) return 0;; esac; return 1; }
mkdir -p "$repo/$out"; cd /tmp; export TMPDIR=/tmp
want bench && { timeout 1500 python3 "$repo/bench.py" --steps 5 --warmup 1 > "$repo/$out/bench.json" 2> "$repo/$out/bench.err" ; }
want bench && { timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/stats" -o r -- python3 "$repo/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-caller-shapes > "$repo/$out/stats.log" 2>&1 ; }
want bench && { cp "$repo/$out"/stats/*kernel_stats.csv "$repo/$out/kernel_stats.csv" 2>/dev/null ; }
want bench && { timeout 600 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 "$repo/bench.py" --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline --no-other-modes --no-other-configs > "$repo/$out/bench_torchrun_n1.json" 2> "$repo/$out/bench_torchrun_n1.err" ; }
want f32w && { bash "$repo/tools/pmc_kernel.sh" "$out/pmc_f32w" resblock_f32w -- "$repo/tools/run_resblock.py" 512 f32 2 > "$repo/$out/pmc_f32w.log" 2>&1 ; }
want bf16 && { B=512 bash "$repo/tools/profile_bf16_modes.sh" "$out/bf16_modes" > "$repo/$out/bf16_modes.log" 2>&1 ; }
want f32s && ( cd "$repo"; GRAFT_REPO_ROOT="$repo" bash tools/pmc_f32s_forms.sh 2>&1 | grep -v amdgpu.ids > "$repo/$out/f32s_forms.txt"; timeout 300 python3 tools/time_f32s_forms.py 512 5 8 2>&1 | grep -v amdgpu.ids >> "$repo/$out/f32s_forms.txt" )
want wb && { timeout 300 python3 "$repo/tools/bench_whitebox.py" 10 5 f32 2>&1 | grep -v amdgpu.ids > "$repo/$out/whitebox.txt" ; }
want wb && { timeout 300 python3 "$repo/tools/bench_whitebox.py" 10 5 bf16 2>&1 | grep -v amdgpu.ids > "$repo/$out/whitebox_bf16.txt" ; }
want wb && { timeout 300 python3 "$repo/tools/bench_whitebox.py" 10 5 bf16s 2>&1 | grep -v amdgpu.ids > "$repo/$out/whitebox_bf16s.txt" ; }
want wb && { timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/wbb_stats" -o r -- python3 "$repo/tools/bench_whitebox.py" 10 5 bf16 > "$repo/$out/wbb_stats.log" 2>&1 ; }
want wb && { cp "$repo/$out"/wbb_stats/*kernel_stats.csv "$repo/$out/whitebox_bf16_kernel_stats.csv" 2>/dev/null ; }
want wb && { timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/wb_stats" -o r -- python3 "$repo/tools/bench_whitebox.py" 10 5 f32 > "$repo/$out/wb_stats.log" 2>&1 ; }
want wb && { cp "$repo/$out"/wb_stats/*kernel_stats.csv "$repo/$out/whitebox_kernel_stats.csv" 2>/dev/null ; }
want wb && { bash "$repo/tools/pmc_kernel.sh" "$out/pmc_bwdb_gate" resblock_bwd_gate_fac_bf16 -- "$repo/tools/bench_whitebox.py" 10 1 bf16 > "$repo/$out/pmc_bwdb_gate.log" 2>&1 ; }
want wb && { bash "$repo/tools/pmc_kernel.sh" "$out/pmc_bwdb_conv" resblock_bwd_conv_bf16 -- "$repo/tools/bench_whitebox.py" 10 1 bf16 > "$repo/$out/pmc_bwdb_conv.log" 2>&1 ; }
want cfg4 && { timeout 600 python3 "$repo/tools/conv_by_shape.py" 256 2>&1 | grep -v amdgpu.ids > "$repo/$out/cfg4_conv_by_shape.txt" ; }
want cfg4 && { timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$repo/$out/cfg4_stats" -o r -- python3 "$repo/tools/run_cfg4_step.py" 256 3 > "$repo/$out/cfg4_step.log" 2>&1 ; }
want cfg4 && { cp "$repo/$out"/cfg4_stats/*kernel_stats.csv "$repo/$out/cfg4_kernel_stats.csv" 2>/dev/null ; }
want adv && { timeout 300 python3 "$repo/tools/adversarial_error.py" f32d f32 f32s f32sw 2>&1 | grep -v amdgpu.ids > "$repo/$out/adversarial_error.txt" ; }
# what travels back: summaries and the small per-dispatch tables only
find "$repo/$out" -name '*.csv' -size +8M -delete
find "$repo/$out" -name '*kernel_trace.csv' -size +2M -delete
ls "$repo/$out"
