#!/bin/bash
# One parameterised GPU-box script (replaces the per-experiment r3*.sh files):  gpurun -- 'bash tools/gpu/run.sh STEP [STEP ...]'
# Every step writes under gpurun_out/<tag>/ (tag = $TAG or "r4").
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
bench_line() { python - "$1" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[1], "steps/s %.3f ms %.2f chk %.15g frac %.3f" % (d["value"], d["ms_per_step"], d["checksum_b_c_l2"], r["frac"]))
    print("   ", {k: (v["avg_ms"], v["GBps"]) for k, v in d["kernels"].items()})
    print("   parity", d.get("parity"))
except Exception as e:
    print(sys.argv[1], "unreadable:", e)
PY
}
for step in "$@"; do
  case $step in
    tests)      python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1; tail -3 $OUT/pytest.txt ;;
    tests-solve) python -m pytest tests/test_gpu_pencil.py tests/test_gpu_reference_pencils.py tests/test_gpu_ivp.py -x -q -m gpu > $OUT/pytest_solve.txt 2>&1; tail -5 $OUT/pytest_solve.txt ;;
    smoke)      python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 ;;
    bench)      python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; bench_line $OUT/bench.json ;;
    bench-quick) python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_quick.json 2> $OUT/bench_quick.err; bench_line $OUT/bench_quick.json ;;
    bench-nosplit) DDH_NO_SPLIT=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_nosplit.json 2> $OUT/bench_nosplit.err; bench_line $OUT/bench_nosplit.json ;;
    bench-cfl)  python bench.py --steps 10 --warmup 3 --cfl --no-cpu-baseline > $OUT/bench_cfl.json 2> $OUT/bench_cfl.err; bench_line $OUT/bench_cfl.json ;;
    profile)    timeout 900 bash tools/profile_round.sh $TAG > $OUT/profile.log 2>&1; tail -3 $OUT/profile.log ;;
    ellband)    # shell LHS at config H: band LU against the dense inverses (solve / factorization device time, per-kernel table)
                export TMPDIR=/tmp
                timeout 200 rocprofv3 --kernel-trace -d /tmp/prof_eb -o eb -- python tools/gpu/ellband_time.py > $OUT/ellband_band.log 2>&1
                grep -E "^(build|plan|solve|factor)" $OUT/ellband_band.log; python tools/gpu/db_stats.py /tmp/prof_eb/eb_results.db 4 | tee $OUT/ellband_kernels.txt
                DDH_SHELL_DENSE=1 timeout 200 python tools/gpu/ellband_time.py 2>&1 | grep -E "^(build|solve|factor)" | tee $OUT/ellband_dense.log ;;
    fused)      # the fused y stage alone: first generation, second generation (LDS-DMA staging default / register loads / twiddles in LDS)
                for envs in "DDH_GW_V2=0" "DDH_GW_V2=1" "DDH_GW_DMA=0" "DDH_GW_DMA=2" "DDH_GW_LPW=4"; do
                  echo "$envs" | tee -a $OUT/fused.txt
                  env $envs FUSED_DERIV=1 python tools/bench_fused.py 2>&1 | tail -1 | tee -a $OUT/fused.txt
                done ;;
    fused-stress) DDH_GW_V2=0 python tools/gpu/fused_stress.py write /tmp/fused_ref.pt 2>&1 | tail -2 | tee $OUT/fused_stress.txt
                python tools/gpu/fused_stress.py check /tmp/fused_ref.pt 2>&1 | tail -3 | tee -a $OUT/fused_stress.txt
                STRESS_REPS=40 FUSED_ZDIV=1 python tools/gpu/fused_stress.py check /tmp/fused_ref.pt 2>&1 | tail -1 | tee -a $OUT/fused_stress.txt ;;
    refactor)   python tools/gpu/refactor_time.py 2>&1 | tail -1 | tee -a $OUT/refactor.txt ;;
    r2)         python -m pytest tests/test_gpu_ivp.py tests/test_gpu_baseline_sizes.py tests/test_gpu_examples.py -x -q -m gpu > $OUT/pytest_r2.txt 2>&1; tail -3 $OUT/pytest_r2.txt
                python tools/bench_configs.py r2 2>&1 | grep -v "^\[" | tee $OUT/r2.txt
                DDH_BLOCK_INVERSE=0 python tools/bench_configs.py r2 2>&1 | grep -v "^\[" | head -2 | tee -a $OUT/r2.txt ;;
    shell-sweep) for envs in "X=0" "DDH_FFT_B=4" "DDH_FFT_B=8" "DDH_FFT_B=16" "DDH_FFT_OCC4=1" "DDH_FFT_B=8 DDH_FFT_OCC4=1"; do echo "$envs" | tee -a $OUT/shell_sweep.txt; env $envs python tools/bench_configs.py shell 2>&1 | grep "steps/s" | head -1 | tee -a $OUT/shell_sweep.txt; done ;;
    tests-fused) python -m pytest tests/test_gpu_wave_transforms.py tests/test_gpu_transforms.py -x -q -m gpu -k "fused" > $OUT/pytest_fused.txt 2>&1; tail -3 $OUT/pytest_fused.txt ;;
    bench-v1)   DDH_GW_V2=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-cfl > $OUT/bench_v1.json 2> $OUT/bench_v1.err; bench_line $OUT/bench_v1.json ;;
    sphere)     python -m pytest tests/test_gpu_swsh.py tests/test_gpu_sphere.py tests/test_gpu_shell.py -x -q -m gpu > $OUT/pytest_sphere.txt 2>&1; tail -3 $OUT/pytest_sphere.txt
                python tools/bench_configs.py sphere 2>&1 | grep -v "^\[" | tail -3 | tee -a $OUT/sphere.txt ;;
    configs)    python tools/bench_configs.py --json > $OUT/configs.json 2> $OUT/configs.err; tail -5 $OUT/configs.json ;;
    offsize)    python tools/bench_configs.py --offsize > $OUT/offsize.json 2> $OUT/offsize.err; python tools/show_offsize.py $OUT/offsize.json ;;
    shares)     # per-rank shares of the strong-scaled problem on one GPU, sweep variants A/B (profiles/*_strong_scaling_shares.txt)
                for sz in 256,512,256 128,512,256 64,512,256; do
                  for var in "default:" "per-thread:DDH_SOLVE_COOP=0" "per-thread-unsplit:DDH_SOLVE_COOP=0 DDH_SPLIT_THREADS=0" "coop-fwd+cb4:DDH_COOP_FWD=1 DDH_COOP_CB=4" "cb4:DDH_COOP_FWD=0 DDH_COOP_CB=4"; do
                    name=${var%%:*}; envs=${var#*:}
                    env $envs python bench.py --size $sz --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-parity --no-cfl > $OUT/share_${sz//,/x}_$name.json 2>/dev/null
                    python - $OUT/share_${sz//,/x}_$name.json "$sz $name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels"]
    print("%-34s ms/step %6.2f  solve %5.2f  fused-y %5.2f  others %5.2f" % (sys.argv[2], d["ms_per_step"], k["pencil_solve"]["avg_ms"],
          k.get("rfft_bilinear_fused", {}).get("avg_ms", 0.0), d["ms_per_step"] - 2 * k["pencil_solve"]["avg_ms"] - 2 * k.get("rfft_bilinear_fused", {}).get("avg_ms", 0.0)))
except Exception as e:
    print(sys.argv[2], "failed:", e)
PY
                  done
                done ;;
    ring-ab)    # backward sweep: direct loads against the LDS-DMA ring, depth 2 / 3 / 4 (DDH_BWD_RING), same box
                for r in 0 2 3 4; do
                  DDH_BWD_RING=$r python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-cfl > $OUT/bench_ring$r.json 2> $OUT/bench_ring$r.err
                  echo "DDH_BWD_RING=$r"; bench_line $OUT/bench_ring$r.json
                done ;;
    emu)        # one rank of the P-rank run on this GPU, loop-back exchange (profiles/r6_rank_emulation.txt)
                python tools/rank_emulation.py --ranks 2,4,8 --rank 1 --steps 10 --warmup 3 ${ONE_GPU_MS:+--single-gpu-ms $ONE_GPU_MS} > $OUT/rank_emulation.jsonl 2> $OUT/rank_emulation.txt; cat $OUT/rank_emulation.txt ;;
    emu-final)  # final table: windows 1 / 2 with and without the emulated wire, P = 2, 4, 8
                for w in 1 2; do for link in 75 0; do
                  echo "DDH_A2A_WINDOWS=$w link $link"; DDH_A2A_WINDOWS=$w python tools/rank_emulation.py --ranks 2,4,8 --rank 1 --steps 10 --warmup 8 --link-gbps $link 2>&1 >> $OUT/rank_emulation_final.jsonl | cut -c1-330
                done; done 2>&1 | tee $OUT/rank_emulation_final.txt ;;
    emu-wire)   # the same with the loop-back exchanges followed by an emulated wire time at 75 GB/s per link: how much of it is hidden
                python tools/rank_emulation.py --ranks 2,4,8 --rank 1 --steps 10 --warmup 8 --link-gbps 75 > $OUT/rank_emulation_wire.jsonl 2> $OUT/rank_emulation_wire.txt; cat $OUT/rank_emulation_wire.txt
                python tools/rank_emulation.py --ranks 2,4,8 --rank 1 --steps 10 --warmup 8 > $OUT/rank_emulation.jsonl 2> $OUT/rank_emulation.txt; cat $OUT/rank_emulation.txt ;;
    emu-defer)  # deferred waits / component-split x steps / z-step prefetch around the blocked exchange, under an emulated wire
                python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $OUT/pytest_multirank.txt 2>&1; tail -3 $OUT/pytest_multirank.txt
                for P in 8 4; do for v in "DDH_A2A_DEFER=0" "DDH_X=1" "DDH_A2A_SPLIT_X=1" "DDH_A2A_PREFETCH=1" "DDH_A2A_PREFETCH=1 DDH_A2A_SPLIT_X=1"; do
                  echo "P=$P $v"; env $v python tools/rank_emulation.py --ranks $P --rank 1 --steps 10 --warmup 8 --link-gbps 75 2>&1 >/dev/null | cut -c1-120; done; done 2>&1 | tee $OUT/emu_defer_ab.txt ;;
    xtiled)     # tile-major state vector: bench A/B with the parity probe, then the suites that step problems with it forced on
                python tools/gpu/xtile_diag.py
                for v in "0" "1" "2" "0" "1" "2"; do env DDH_X_TILED=$v python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-cfl > $OUT/bench_xtiled.json 2> $OUT/bench_xtiled.err
                  echo "DDH_X_TILED=$v"; bench_line $OUT/bench_xtiled.json | cut -c1-400
                  python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('   parity', {k: d['parity'][k] for k in ('max_residual', 'max_solution_error')})" $OUT/bench_xtiled.json
                done
                DDH_X_TILED_MIN=0 DDH_RHS_TILING_MIN=0 python -m pytest tests/test_gpu_ivp.py tests/test_gpu_baseline_sizes.py tests/test_gpu_reference_pencils.py tests/test_gpu_examples.py tests/test_gpu_state_tiling.py -x -q -m gpu > $OUT/pytest_xtiled.txt 2>&1; tail -5 $OUT/pytest_xtiled.txt ;;
    emu-windows) # the grid stage in windows of z planes pipelined against windowed exchanges, under an emulated wire
                python -m pytest tests/test_gpu_multirank.py -x -q -m gpu > $OUT/pytest_multirank.txt 2>&1; tail -3 $OUT/pytest_multirank.txt
                for P in 8 4; do for v in 1 2 4; do
                  echo "P=$P DDH_A2A_WINDOWS=$v"; env DDH_A2A_WINDOWS=$v python tools/rank_emulation.py --ranks $P --rank 1 --steps 10 --warmup 8 --link-gbps 75 2>&1 >/dev/null | cut -c1-220; done; done 2>&1 | tee $OUT/emu_windows_ab.txt ;;
    bwd-rowmajor) # timing experiment: the backward sweep reading the factor rows as if stored row-major over the blocks
                for v in 0 128; do DDH_BWD_DBG=$v python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-cfl --no-parity > $OUT/bench_bwddbg$v.json 2> $OUT/bench_bwddbg$v.err
                  echo "DDH_BWD_DBG=$v"; bench_line $OUT/bench_bwddbg$v.json; done ;;
    deep-ab)    # few-system sweeps: deep register prefetch (solve_*_deep_kernel) against the plain kernels, bit identity + time
                for sz in 64,512,256 128,512,256; do for v in 0 1 "1 DDH_BWD_DEEP_PD=4"; do
                  env DDH_SWEEP_DEEP=$v python bench.py --size $sz --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-cfl --no-parity > $OUT/bench_deep.json 2> $OUT/bench_deep.err
                  echo "size $sz DDH_SWEEP_DEEP=$v"; bench_line $OUT/bench_deep.json; done; done
                DDH_SWEEP_DEEP=1 python -m pytest tests/test_gpu_pencil.py tests/test_gpu_reference_pencils.py -x -q -m gpu > $OUT/pytest_deep.txt 2>&1; tail -3 $OUT/pytest_deep.txt ;;
    deep-big)   # the deep sweeps beyond their default range: 2 and 4 waves per SIMD worth of threads
                for sz in 256,512,256 512,512,256; do for v in 0 1 "1 DDH_BWD_DEEP_PD=4"; do
                  env DDH_SWEEP_DEEP=$v python bench.py --size $sz --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-cfl --no-parity > $OUT/bench_deep.json 2> $OUT/bench_deep.err
                  echo "size $sz DDH_SWEEP_DEEP=$v"; bench_line $OUT/bench_deep.json; done; done ;;
    rowfill-ab) # backward sweep: entry pairs beyond the measured fill of a U row not loaded (LuDev::wrow), against all pairs
                for v in 0 1 0 1; do DDH_BWD_ROW_FILL=$v python bench.py --steps 10 --warmup 3 --repeats 1 --no-cpu-baseline --no-cfl > $OUT/bench_rowfill$v.json 2> $OUT/bench_rowfill$v.err
                  echo "DDH_BWD_ROW_FILL=$v"; bench_line $OUT/bench_rowfill$v.json; done ;;
    emu-block)  # the deep sweeps' workgroup size at the P = 4 / 8 shares
                for b in 256 128 64; do echo "DDH_DEEP_BLOCK=$b"; DDH_DEEP_BLOCK=$b python tools/rank_emulation.py --ranks 4,8 --rank 1 --steps 10 --warmup 3 2>&1 >/dev/null | cut -c1-150; done ;;
    emu-ring)   # the P = 4 / 8 shares with the backward sweep's LDS-DMA ring, depth 2 / 3 / 4
                for r in 2 3 4; do echo "DDH_BWD_RING=$r" | tee -a $OUT/rank_emulation_ring.txt
                  DDH_BWD_RING=$r python tools/rank_emulation.py --ranks 4,8 --rank 1 --steps 10 --warmup 3 2>&1 >/dev/null | tee -a $OUT/rank_emulation_ring.txt; done ;;
    tests-new)  python -m pytest tests/test_gpu_wave_transforms.py tests/test_gpu_comm.py tests/test_gpu_pencil.py tests/test_gpu_reference_pencils.py tests/test_gpu_baseline_sizes.py -x -q -m gpu -s > $OUT/pytest_new.txt 2>&1; tail -5 $OUT/pytest_new.txt; grep "end state vs" $OUT/pytest_new.txt ;;
    *)          echo "running: $step"; bash -c "$step" ;;
  esac
done
