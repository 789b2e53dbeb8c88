# usage (GPU box): bash tools/collect_round.sh [round = r06]  -- every table of a round into gpurun_out/<round>/ (tools/publish_profiles.sh copies them to profiles/)
set -u
R=${1:-r06}
O=gpurun_out/$R
mkdir -p $O
# the tables below are those of rounds 1-5: problems started in the order given (bench.py sets the order of each of its legs itself);
# the library's default order gets a table of its own at the end
export MPCGPU_ORDER=as_given
python bench.py --steps 5 --warmup 1 > $O/bench_line.json 2> $O/bench.err
(echo "# tools/phase_prof.py on a -DMPC_PROFILE build of the final source of this round (throughput kernel; shader-clock cycles per phase; 16 resident wavefronts per CU at N = 20, 12 at N = 40)"; MPCGPU_LIB=$PWD/build_ab/libmpcgpu_prof.so python tools/phase_prof.py 8192 20; MPCGPU_LIB=$PWD/build_ab/libmpcgpu_prof.so python tools/phase_prof.py 4096 40) > $O/phase_table.txt 2>&1
(echo "# config 2: N_hor = 20, 4 dynamic obstacles, B = 1024 (tools/prof_solve.py 1024 3 4 20)"; python tools/prof_solve.py 1024 3 4 20; echo "# config 3: N_hor = 40, 8 dynamic obstacles, B = 4096 / 16384 (tools/prof_solve.py B 3 8 40)"; python tools/prof_solve.py 4096 3 8 40; python tools/prof_solve.py 16384 2 8 40) > $O/config2_config3.txt 2>&1
(echo "# tools/prof_solve.py B 2 8 20: benchmark scene family, N_hor = 20, 8 dynamic obstacles, cold start, solve kernel ms"; for B in 8192 16384 32768 65536 131072; do python tools/prof_solve.py $B 2 8 20; done) > $O/batch_scaling.txt 2>&1
(echo "# Gram form vs two-loop recursion of the L-BFGS operator, same source otherwise (tools/ab.sh; kernel ms, 2 runs each)"; for B in 32768 8192; do echo "## N_hor = 20, B = $B"; ABB=$B bash tools/ab.sh trajtrack_mpcndqn_rlboost_amd/libmpcgpu.so trajtrack_mpcndqn_rlboost_amd/variants/libmpcgpu_twoloop.so; done) > $O/lbfgs_gram_ab.txt 2>&1
python tools/latency.py 20 > $O/latency.txt 2>&1
(echo "# tools/closed_loop.py 8192 30 4 [warm]: DeviceTracker, scene 1 with 4 constant-velocity discs, 30 timed ticks after 5 warm-up ticks, HIP-event wall time per tick"; python tools/closed_loop.py 8192 30 4; python tools/closed_loop.py 8192 30 4 warm) > $O/closed_loop.txt 2>&1
python tests/tools/kkt_report.py > $O/kkt_report.txt 2>&1
python tests/tools/parity_report.py > $O/parity_report.txt 2>&1
(echo "# balanced walk of the dynamic rows at N_hor = 40 (product) against the row walk of rounds 1-3 (variants/libmpcgpu_rowwalk40.so): tools/quick_ab.py B 2 40 bench"; for B in 16384 4096; do python tools/quick_ab.py $B 2 40 bench; MPCGPU_LIB=trajtrack_mpcndqn_rlboost_amd/variants/libmpcgpu_rowwalk40.so python tools/quick_ab.py $B 2 40 bench; done) 2>&1 | grep -v amdgpu.ids > $O/balanced_walk_ab.txt
bash tools/n40_counters.sh > $O/counters_N40.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_hybrid.py -q -m gpu -s -k "configured or device" > $O/hybrid.txt 2>&1
python tools/scanner_replay.py 256 > $O/scanner_replay_B256.txt 2>&1
python tools/train_dqn.py --envs 4096 --timesteps 401408 --graph --save $O/dqn_ckpt > $O/dqn_config5.txt 2>&1
ls $O/dqn_ckpt >> $O/dqn_config5.txt; rm -rf $O/dqn_ckpt
bash tools/collect_profiles.sh gpurun_out/raw_$R > $O/collect.log 2>&1
bash tools/pmc_stalls.sh 8192 gpurun_out/pmc_stalls_$R > $O/pmc_stalls_B8192.txt 2>&1
(echo "# dispatch order (MPCGPU_OPT_ORDER): tools/prof_solve.py B 4 8 N, kernel ms of four consecutive calls of one handle (the first has no hints)"; for a in "8192 4 8 20" "32768 4 8 20" "4096 4 8 40" "16384 4 8 40"; do for o in as_given longest_first; do echo "## $a  MPCGPU_ORDER=$o"; MPCGPU_ORDER=$o python tools/prof_solve.py $a 2>&1 | grep solve_ms | sed "s/inner.*//"; done; done) > $O/order_table.txt 2>&1
(echo "# tail promotion off (K = 0) / on with the continuation on the side stream while the launch drains (K = -1: the library default)"; python tools/tail_ab.py 20 4096,8192,32768,131072 0,-1 2; python tools/tail_ab.py 40 4096,16384 0,-1 2; python tools/tail_ab.py 20 8192 0,-1 2 avoidance; python tools/tail_ab.py 20 8192 0,-1 2 passing; echo "# ... with the continuation as the launch BEHIND the throughput kernel (MPCGPU_TAIL_CONCURRENT=0: what a captured call records)"; MPCGPU_TAIL_CONCURRENT=0 python tools/tail_ab.py 20 4096,8192,32768 -1 2; MPCGPU_TAIL_CONCURRENT=0 python tools/tail_ab.py 40 4096,16384 -1 2) 2>&1 | grep -v amdgpu.ids > $O/tail_promotion_final.txt
(python tools/closed_loop.py 8192 30 4 cold capacity) 2>&1 | grep -v amdgpu.ids > $O/realtime_capacity.txt
(python tools/closed_loop.py 8192 30 4 cold streams) 2>&1 | grep -v amdgpu.ids > $O/closed_loop_streams.txt
# round 6: the two readings of the penalty-stall rule on every workload; the concurrent continuation beside foreign work; the environment kernel
python tools/stall_rule_report.py 2>&1 | grep -v amdgpu.ids > $O/stall_rule.txt
python tools/foreign_work_soak.py --calls 200 2>&1 | grep -v amdgpu.ids > $O/foreign_work.txt
python tools/bench_env.py 2>&1 | grep -v amdgpu.ids > $O/env_bench.txt
python tools/team_sweep.py bench either > $O/team_sweep.txt 2>&1; for a in "passing either" "avoidance either" "bench both" "passing both" "avoidance both" "bench either 40" "passing either 40"; do python tools/team_sweep.py $a 2>&1 | grep -v amdgpu.ids >> $O/team_sweep.txt; done
(echo "# tests/tools/fuzz_parity.py: for s in 0..5: python tests/tools/fuzz_parity.py 150 \$s 12 (per-trial lines dropped)"; for s in 0 1 2 3 4 5; do echo "== seed $s"; python tests/tools/fuzz_parity.py 150 $s 12 2>&1 | grep -v "^trial \|^full  *[0-9]\|amdgpu.ids"; done) > $O/fuzz_parity.txt 2>&1
python __graft_entry__.py smoke > $O/smoke.txt 2>&1
tail -2 $O/bench.err; cat $O/bench_line.json | cut -c1-400
