#!/bin/bash
# Lab notebook of rounds 3-4 as ONE runner: every GPU call of those rounds is a function below (body = the one-shot script it
# used to be: scripts/experiments/round{3,4}/gpu_*.sh, 86 files until round 5); `call <name>` runs one, `list` prints them with
# their first comment line.  Many of them need experiment builds (scripts/build_variant.sh, hot_ablation.py) or patches that were
# applied to a temporary copy of the sources at the time; they are kept as the record of WHAT was measured (results:
# profiles/r03_probes.txt, profiles/r04_probes.txt, HISTORY.md), not as a regression suite.
#   bash scripts/experiments/notebook_r3_r4.sh list
#   /usr/local/graft/bin/gpurun -- bash scripts/experiments/notebook_r3_r4.sh call r4t
set -u

r3_coldpaths() { ( # ---- round3/gpu_coldpaths.sh
# cold-protocol figures of the small configs under each path (which path should the auto rule pick when the working set is cold?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/r3_coldpaths.txt
: > $out
run() { echo "## $*" >> $out; timeout 300 python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line); r=d.get("roofline",{})
    print("   cold_us=%s frac=%s warm=%s slabs=%s" % (r.get("launch_us"), r.get("frac"), (r.get("warm") or {}).get("launch_us"), d.get("config",{}).get("column_slabs")))
' >> $out; }
for w in webbase scircuit; do
  run --workload $w
  run --workload $w --slabs 0
  run --workload $w --slabs 0 --x-window off
  run --workload $w --slabs 2
  run --workload $w --slabs 8
  run --workload $w --mode two-pass --slabs 0
done
run --workload nd24k --dtype f32
run --workload nd24k --dtype f32 --x-window off
run --workload nd24k --dtype f32 --sigma 8
run --workload nd24k --dtype f32 --sigma 32
cat $out
) }

r3_final() { ( # ---- round3/gpu_final.sh
# last call of the round: full -m gpu suite, smoke, then the round's profiles and the default bench line on the final sources
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4) > gpurun_out/final_tests.txt 2>&1
cat gpurun_out/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/final_smoke.txt
bash scripts/experiments/round3/gpu_r3s.sh
) }

r3a() { ( # ---- round3/gpu_r3a.sh
# round 3, call A: correctness of the range kernel, then same-call A/B against the round-2 library
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3a_tests.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat" 2>&1 | tail -15 >> gpurun_out/r3a_tests.txt
cat gpurun_out/r3a_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" scripts/probes/libcsr5hip_prev.so benchmark_spmv_using_csr5_amd/libcsr5hip.so 2>&1 | tee gpurun_out/r3a_ab.txt
for s in 16 32; do
  echo "== slabs $s"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs $s
done 2>&1 | tee gpurun_out/r3a_kstats.txt
) }

r3b() { ( # ---- round3/gpu_r3b.sh
# round 3, call B: parallel finish kernel + run-streaming combine: tests, per-kernel times by slab count, all-hot floor
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3b_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled" 2>&1 | tail -6 >> gpurun_out/r3b_tests.txt
cat gpurun_out/r3b_tests.txt
for s in 16 32 64; do
  echo "== rmat24 slabs $s"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs $s
done 2>&1 | tee gpurun_out/r3b_kstats.txt
for s in 8 16; do
  echo "== rmat22 slabs $s"; bash scripts/gpu_kstats.sh --workload rmat22 --slabs $s
done 2>&1 | tee -a gpurun_out/r3b_kstats.txt
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase 2>&1 | tee -a gpurun_out/r3b_kstats.txt
timeout 600 python scripts/experiments/hot_floor.py --scale 24 --hubs 131072 2>&1 | tee gpurun_out/r3b_hot_floor.txt
) }

r3d() { ( # ---- round3/gpu_r3d.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(cd scripts/probes && timeout 300 ./stream_shape) > gpurun_out/r3d_stream_shape.txt 2>&1
cat gpurun_out/r3d_stream_shape.txt
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r3d_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled" 2>&1 | tail -4 >> gpurun_out/r3d_tests.txt
cat gpurun_out/r3d_tests.txt
for s in 16 32; do
  echo "== rmat24 slabs $s"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs $s
done 2>&1 | tee gpurun_out/r3d_kstats.txt
echo "== rmat22"; bash scripts/gpu_kstats.sh --workload rmat22 2>&1 | tee -a gpurun_out/r3d_kstats.txt
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase 2>&1 | tee -a gpurun_out/r3d_kstats.txt
) }

r3e() { ( # ---- round3/gpu_r3e.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3e_tests.txt
cat gpurun_out/r3e_tests.txt
{
echo "== rmat24 default"; bash scripts/gpu_kstats.sh --workload rmat24
echo "== rmat24 zero-empty"; bash scripts/gpu_kstats.sh --workload rmat24 --zero-empty 1
echo "== rmat24 32 slabs"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs 32
echo "== rmat22"; bash scripts/gpu_kstats.sh --workload rmat22
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase
} 2>&1 | tee gpurun_out/r3e_kstats.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_depth1.so scripts/probes/libcsr5hip_cg1024.so scripts/probes/libcsr5hip_cg4096.so 2>&1 | tee gpurun_out/r3e_ab.txt
) }

r3f() { ( # ---- round3/gpu_r3f.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3f_tests.txt
cat gpurun_out/r3f_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_cb64.so scripts/probes/libcsr5hip_cb128.so scripts/probes/libcsr5hip_cb512.so scripts/probes/libcsr5hip_cb1024.so scripts/probes/libcsr5hip_cr128.so 2>&1 | tee gpurun_out/r3f_ab.txt
) }

r3h() { ( # ---- round3/gpu_r3h.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "" abl_nocold abl_nogather abl_nostore abl_notable; do
  lib=benchmark_spmv_using_csr5_amd/libcsr5hip.so; [ -n "$v" ] && lib=scripts/probes/libcsr5hip_$v.so
  echo "== ${v:-product}"; CSR5HIP_LIB=$PWD/$lib bash scripts/gpu_kstats.sh --workload rmat24
done 2>&1 | tee gpurun_out/r3h_ablation.txt
) }

r3j() { ( # ---- round3/gpu_r3j.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3j_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled" 2>&1 | tail -3 >> gpurun_out/r3j_tests.txt
cat gpurun_out/r3j_tests.txt
{
echo "== rmat24 16"; bash scripts/gpu_kstats.sh --workload rmat24
echo "== rmat24 32 slabs"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs 32
echo "== rmat22"; bash scripts/gpu_kstats.sh --workload rmat22
echo "== rmat22 16"; bash scripts/gpu_kstats.sh --workload rmat22 --slabs 16
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase
} 2>&1 | tee gpurun_out/r3j_kstats.txt
) }

r3k() { ( # ---- round3/gpu_r3k.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_abl_ntstore.so 2>&1 | tee gpurun_out/r3k_ab.txt
) }

r3l() { ( # ---- round3/gpu_r3l.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_goldens.py tests/test_gpu_full_size.py tests/test_gpu_slabs.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3l_tests.txt
timeout 1500 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -25 >> gpurun_out/r3l_tests.txt
cat gpurun_out/r3l_tests.txt
) }

r3n() { ( # ---- round3/gpu_r3n.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r3n_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled or checkpoint" 2>&1 | tail -4 >> gpurun_out/r3n_tests.txt
cat gpurun_out/r3n_tests.txt
bash scripts/gpu_convtrace.sh rmat24 52 2>&1 | tee gpurun_out/r3n_convtrace.txt
) }

r3o() { ( # ---- round3/gpu_r3o.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in pieces2 pieces4; do
  CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so timeout 600 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu -k "hot" 2>&1 | tail -1
done | tee gpurun_out/r3o_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_pieces2.so scripts/probes/libcsr5hip_pieces4.so scripts/probes/libcsr5hip_pieces8.so 2>&1 | tee gpurun_out/r3o_ab.txt
) }

r3p() { ( # ---- round3/gpu_r3p.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_abl_gather_nt.so scripts/probes/libcsr5hip_abl_gather_sc1.so scripts/probes/libcsr5hip_abl_gather_sc0sc1.so 2>&1 | tee gpurun_out/r3p_ab.txt
) }

r3q() { ( # ---- round3/gpu_r3q.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in seg256 seg128; do
  CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_parity.py -x -q -m gpu -k "hot or fuzz or rmat" 2>&1 | tail -1
done | tee gpurun_out/r3q_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_seg256.so scripts/probes/libcsr5hip_seg128.so 2>&1 | tee gpurun_out/r3q_ab.txt
) }

r3r() { ( # ---- round3/gpu_r3r.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r3r_full_gpu_suite.txt 2>&1
cat gpurun_out/r3r_full_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r3r_smoke.txt
) }

r3s() { ( # ---- round3/gpu_r3s.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ROUND=r03 bash scripts/gpu_profiles_round.sh > gpurun_out/r3s_profiles.log 2>&1
tail -12 gpurun_out/r3s_profiles.log
(time python bench.py) > gpurun_out/r3s_bench_default.txt 2>&1
tail -c 3000 gpurun_out/r3s_bench_default.txt
) }

r3t() { ( # ---- round3/gpu_r3t.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for w in nd24k scircuit; do
  for rk in off force; do
    echo "== $w range-kernel $rk"
    timeout 300 python bench.py --workload $w --no-cpu-baseline --range-kernel $rk 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('  range_kernel', d['config']['range_kernel'], 'cold us', r['launch_us'], 'frac', r['frac'], '| warm us', r['warm']['launch_us'], 'frac', r['warm']['frac'])"
  done
done 2>&1 | tee gpurun_out/r3t_range_plain.txt
) }

r3w() { ( # ---- round3/gpu_r3w.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "" abl_rowscan_noflag abl_rowscan_noempty; do
  lib=benchmark_spmv_using_csr5_amd/libcsr5hip.so; [ -n "$v" ] && lib=scripts/probes/libcsr5hip_$v.so
  echo "== ${v:-product}"; CSR5HIP_LIB=$PWD/$lib bash scripts/gpu_convtrace.sh rmat24 52 2>&1 | grep -E "k_row_scan|k_slab_scatter|k_tile_desc|R-MAT" | cut -c1-110
done 2>&1 | tee gpurun_out/r3w_rowscan.txt
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -2 | tee -a gpurun_out/r3w_rowscan.txt
) }

r3x() { ( # ---- round3/gpu_r3x.sh
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r3x_tests.txt 2>&1
cat gpurun_out/r3x_tests.txt
bash scripts/gpu_convtrace.sh rmat24 52 2>&1 | tee gpurun_out/r3x_convtrace.txt | cut -c1-120 | head -52
for w in scircuit webbase nd24k rmat22; do timeout 300 python scripts/bench_convert.py --workload $w 2>&1 | tail -1; done | tee gpurun_out/r3x_convert.txt
) }

r3y() { ( # ---- round3/gpu_r3y.sh
# packed column codes: full GPU suite, A/B against the previous build, conversion trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r3y_tests.txt 2>&1
cat gpurun_out/r3y_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" scripts/probes/libcsr5hip_prev.so benchmark_spmv_using_csr5_amd/libcsr5hip.so 2>&1 | tee gpurun_out/r3y_ab.txt | cut -c1-200
bash scripts/gpu_convtrace.sh rmat24 60 2>&1 | tee gpurun_out/r3y_convtrace.txt | grep -E "GFLOPS|hot_encode|k_transpose|stats_export|spmv_range|combine" | cut -c1-130
) }

r3z() { ( # ---- round3/gpu_r3z.sh
# every row block of the strong-scaling R-MAT 24 ALONE on the one GPU, round-3 kernels (estimate of the N-GPU step: the slowest block)
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "## scripts/experiments/shard_alone.py: every row block of the strong-scaling R-MAT 24 ALONE on one MI355X (cost balance nnz + 2*rows), round-3 kernels"
python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
python scripts/experiments/shard_alone.py --scale 24 --world 2 --ranks 0,1
python scripts/experiments/shard_alone.py --scale 24 --world 4 --ranks 0,1,2,3
python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,1,2,3,4,5,6,7
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r3z_shards.txt
) }

r4a() { ( # ---- round4/gpu_r4a.sh
# round 4, call a: parity of the permuted-x path, then same-call A/B against the round-3 library
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w r03"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_r03.so one --workload $w
    echo "== $w new snapshot"; one --workload $w
    echo "== $w new live"; one --workload $w --x-snapshot 0
  done
done
python bench.py --no-cpu-baseline --no-sub-configs 2>&1 | tail -1 > gpurun_out/r4a_rmat24.json
) }

r4ab() { ( # ---- round4/gpu_r4ab.sh
# round 4, call ab: where does the final tile kernel's time go?  builds of the product with one piece of work removed
# (scripts/experiments/hot_ablation.py; wrong results by design), kernel averages from rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
for v in product nocold nogather nostore notable; do
  lib=$GRAFT_REPO_ROOT/scripts/probes/libcsr5hip_abl_$v.so
  [ $v = product ] && lib=$GRAFT_REPO_ROOT/benchmark_spmv_using_csr5_amd/libcsr5hip.so
  rm -rf /tmp/ks_$v
  CSR5HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 3 > /tmp/ks_$v.log 2>&1
  f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep '"metric"' /tmp/ks_$v.log | tail -1 | python $GRAFT_REPO_ROOT/scripts/benchline.py | cut -c88-130)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_spmv_range", "k_slab_combine", "k_range_finish")):
        print("   %-40s calls %3s avg %8.1f us" % (r["Name"].split("(")[0][5:45], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
) }

r4ac() { ( # ---- round4/gpu_r4ac.sh
# round 4, call ac: child sigma 8 / 12 / 16 at 8 wavefronts per CU (a larger tile amortises the per-tile work that two wavefronts
# per SIMD no longer hide; its y region takes LDS from the table: 16 384 / 14 336 / 12 288 slots), and sigma 16 at 6 wavefronts
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w sigma 8"; one --workload $w
    echo "== $w sigma 12"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s12.so one --workload $w
    echo "== $w sigma 16"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s16.so one --workload $w
    echo "== $w sigma 16, 6 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s16w6.so one --workload $w
  done
done
) }

r4ad() { ( # ---- round4/gpu_r4ad.sh
# round 4, call ad: fp64 child sigma 16 / 12 / 8 with the SAME 16 384-slot table (4-KB y region, a tile of very short rows walks its
# flags in windows), 8 wavefronts per CU; parity first
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w sigma 8"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s8.so one --workload $w
    echo "== $w sigma 12"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s12w.so one --workload $w
    echo "== $w sigma 16"; one --workload $w
  done
done
) }

r4ae() { ( # ---- round4/gpu_r4ae.sh
# round 4, call ae: slab hash granule (columns hashed together) now that cold gathers read the dense permuted copy, not lines of x
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for sh in 4 0 2 8; do echo "== rmat24 shift $sh"; one --workload rmat24 --slab-shift $sh; done
  for sh in 4 0 2 8; do echo "== rmat22 shift $sh"; one --workload rmat22 --slab-shift $sh; done
done
) }

r4af() { ( # ---- round4/gpu_r4af.sh
# round 4, call af: 16 (12) wavefronts per CU taking turns on 8 (6) y-compaction regions (LDS lock): occupancy of the round-3
# geometry with the table of the round-4 one
CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16sh2.so timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16sh2.so CSR5_FUZZ_SEED=77 CSR5_FUZZ_CASES=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 8 waves (product)"; one --workload $w
    echo "== $w 16 waves on 8 regions"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16sh2.so one --workload $w
    echo "== $w 12 waves on 6 regions"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w12sh2.so one --workload $w
  done
done
) }

r4ak() { ( # ---- round4/gpu_r4ak.sh
# round 4, call ak: the child's bit flags ride in bit 22 of the column codes, y_offset is recomputed in the kernel, the descriptor
# array is not read (-256 B of 6 144 B per tile): parity, fuzz, A/B vs the previous library
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
CSR5_FUZZ_SEED=611 CSR5_FUZZ_CASES=3000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
CSR5_FUZZ_SCALE=30 CSR5_FUZZ_SEED=612 CSR5_FUZZ_CASES=600 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2 3; do
  for w in rmat24 rmat22; do
    echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_desc.so one --workload $w
    echo "== $w after"; one --workload $w
  done
done
) }

r4al() { ( # ---- round4/gpu_r4al.sh
# round 4, call al: whole gpu suite after dropping the hot child's descriptor array; conversion time; bench
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
CSR5_FUZZ_SEED=613 CSR5_FUZZ_CASES=3000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for w in rmat24 rmat22; do echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_desc.so one --workload $w; echo "== $w after"; one --workload $w; done
timeout 600 python scripts/experiments/scale_check.py --scale 25 2>&1 | tail -1
) }

r4am() { ( # ---- round4/gpu_r4am.sh
# round 4, call am: k_spmv touches the streams of the tile one resident set ahead (CSR5_PREFETCH = percent of a resident set);
# cold and warm step of the three small configs, each value of the knob twice
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {})
        print('%-28s cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (d['config']['workload'][:28], r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in nd24k webbase scircuit; do
  for p in 0 50 100 200 0 100; do
    echo -n "prefetch $p: "; CSR5_PREFETCH=$p timeout 300 python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line
  done
done
) }

r4an() { ( # ---- round4/gpu_r4an.sh
# round 4, call an: the stream / gather probe again with its modes repaired (round 4's first version computed the column codes of the
# "no streams" and "streams only" modes with 64-bit hashes -- 388 us of arithmetic -- and the compiler dropped the unused stream loads),
# plus wavefront specialisation: producer wavefronts stream into LDS buffers, consumer wavefronts gather and compute
cd scripts/probes
timeout 120 ./lds_dma_streams 268435456 28,0
timeout 120 ./lds_dma_streams_w0 268435456 28,0
) }

r4ao() { ( # ---- round4/gpu_r4ao.sh
# round 4, call ao: the combine as a stream (persistent wavefronts, two row blocks in flight each); CSR5_COMBINE_STREAM = workgroups per CU
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for m in 0 3 2 4 0 3; do echo -n "$w stream=$m: "; CSR5_COMBINE_STREAM=$m one --workload $w; done; done
CSR5_COMBINE_STREAM=3 timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
cd /tmp && export TMPDIR=/tmp
CSR5_COMBINE_STREAM=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
grep -h "combine\|k_spmv_range\|range_finish" $(find /tmp/pc -name "*kernel_stats.csv") | cut -c1-150
) }

r4ap() { ( # ---- round4/gpu_r4ap.sh
# round 4, call ap: the combine's LDS read-modify-write chain (16 dependent ds_read / add / ds_write rounds) as ds_add_f64 without return
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22 webbase; do for m in 0 1 0 1; do echo -n "$w atomic=$m: "; CSR5_COMBINE_ATOMIC=$m one --workload $w --no-cold; done; done
CSR5_COMBINE_ATOMIC=1 timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
cd /tmp && export TMPDIR=/tmp
CSR5_COMBINE_ATOMIC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
grep -h "combine\|k_spmv_range\|range_finish" $(find /tmp/pc -name "*kernel_stats.csv") | cut -c1-60,180-260
) }

r4aq() { ( # ---- round4/gpu_r4aq.sh
# round 4, call aq: what holds k_slab_combine (R-MAT 24)?  LDS conflicts, the dispatcher's resource stalls, instruction classes
mkdir -p gpurun_out
{
export KFILTER="k_slab_combine<double, 16"
PMC="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" bash scripts/gpu_pmc1.sh lds --no-sub-configs
PMC="SPI_RA_LDS_CU_FULL_CSN SPI_RA_TGLIM_CU_FULL_CSN SPI_RA_WAVE_SIMD_FULL_CSN SPI_RA_VGPR_SIMD_FULL_CSN" bash scripts/gpu_pmc1.sh spi --no-sub-configs
PMC="SPI_RA_REQ_NO_ALLOC_CSN SPI_RA_RES_STALL_CSN SPI_CSN_BUSY SPI_CSN_WAVE" bash scripts/gpu_pmc1.sh spi2 --no-sub-configs
PMC="SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" bash scripts/gpu_pmc1.sh cls --no-sub-configs
PMC="SQ_LEVEL_WAVES SQ_BUSY_CYCLES SQ_WAVES SQ_IFETCH" bash scripts/gpu_pmc1.sh lvl --no-sub-configs
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq --no-sub-configs
PMC="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU" bash scripts/gpu_pmc1.sh ins --no-sub-configs
rm -rf gpurun_out/pmc1_*
} 2>&1 | grep -v "^$" > gpurun_out/r04_pmc_combine_raw.txt
cat gpurun_out/r04_pmc_combine_raw.txt
) }

r4ar() { ( # ---- round4/gpu_r4ar.sh
# round 4, call ar: workgroup size of k_slab_combine (one row block per wavefront): 64 / 128 / 256 (product) / 512 / 1024 threads
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22 webbase; do for b in 256 64 128 512 1024 256; do
  lib=$PWD/scripts/probes/libcsr5hip_cb$b.so; [ $b = 256 ] && lib=$PWD/benchmark_spmv_using_csr5_amd/libcsr5hip.so
  echo -n "$w combine block $b: "; CSR5HIP_LIB=$lib one --workload $w --no-cold; done; done
) }

r4as() { ( # ---- round4/gpu_r4as.sh
# round 4, call as: TIMING of k_slab_combine if a row block's S runs of partials lay one behind the other in P (block-major P instead of
# slab-major; results wrong by design: only the addresses change) -- is the combine held by its 16 scattered short runs per block?
cd /tmp && export TMPDIR=/tmp
for m in 0 2 0 2; do
  rm -rf /tmp/pc; CSR5_COMBINE_ATOMIC=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
  echo "mode $m:"; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done
for m in 0 2; do
  rm -rf /tmp/pc; CSR5_COMBINE_ATOMIC=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --workload rmat22 --steps 20 --warmup 5 > /dev/null 2>&1
  echo "rmat22 mode $m:"; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done
) }

r4at() { ( # ---- round4/gpu_r4at.sh
# round 4, call at: ablations of k_slab_combine on R-MAT 24 (wrong results by design): 3 = no partial / row-byte loads, 6 = no partial
# loads, 7 = no row-byte loads, 4 = no stores, 5 = no LDS accumulation, 0 = product
cd /tmp && export TMPDIR=/tmp
for m in 0 3 6 7 4 5 0; do
  rm -rf /tmp/pc; CSR5_COMBINE_ATOMIC=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
  echo -n "mode $m: "; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done
) }

r4au() { ( # ---- round4/gpu_r4au.sh
# round 4, call au: the combine as a stream, second form (unit of work = one round of a block, next round's loads issued first, ds_add_f64)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for m in 0 1 0 1; do echo -n "$w stream=$m: "; CSR5_COMBINE_STREAM=$m one --workload $w; done; done
CSR5_COMBINE_STREAM=1 timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
cd /tmp && export TMPDIR=/tmp
CSR5_COMBINE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
) }

r4av() { ( # ---- round4/gpu_r4av.sh
# round 4, call av: k_slab_combine with 32-bit byte offsets on a scalar base (288 -> 256 vector instructions per block)
cd /tmp && export TMPDIR=/tmp
for v in product off32 product off32; do
  lib=$GRAFT_REPO_ROOT/scripts/probes/libcsr5hip_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/benchmark_spmv_using_csr5_amd/libcsr5hip.so
  rm -rf /tmp/pc; CSR5HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
  echo -n "$v: "; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done
) }

r4aw() { ( # ---- round4/gpu_r4aw.sh
# round 4, call aw: per-kernel times of row blocks 1 and 3 of 8 (R-MAT 24) alone on the GPU: where do the blocks' 170-178 us go?
cd /tmp && export TMPDIR=/tmp
for r in 1 3 7; do
  rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks $r 2>/dev/null | grep '"rank"' | cut -c1-170
  grep -h "k_spmv_range\|k_slab_combine\|k_range_finish\|k_x_permute" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/^"void csr5::\([a-z_]*\).*)",/\1 /' | cut -c1-90
done
) }

r4ax() { ( # ---- round4/gpu_r4ax.sh
# round 4, call ax: gaps between the three kernels of a step inside the replayed graph (row block 3 of 8, and the whole matrix)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 3 2>/dev/null | grep '"rank"' | cut -c1-150
python $GRAFT_REPO_ROOT/scripts/experiments/kernel_gaps.py $(find /tmp/pc -name "*kernel_trace.csv") | grep "range\|combine\|finish"
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 2>/dev/null | tail -1 | python $GRAFT_REPO_ROOT/scripts/benchline.py | cut -c60-140
python $GRAFT_REPO_ROOT/scripts/experiments/kernel_gaps.py $(find /tmp/pc -name "*kernel_trace.csv") | grep "range\|combine\|finish"
) }

r4ay() { ( # ---- round4/gpu_r4ay.sh
# round 4, call ay: CSR5HIP_OPT_NARROW_VALUES (fp32-exact fp64 values streamed as fp32): parity and the side figure of the bench
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for w in rmat24 rmat22; do python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['config']['workload'][:20], d['value'], r['launch_us'], r['frac'], json.dumps(r.get('narrowed_values'))[:400])"; done
) }

r4az() { ( # ---- round4/gpu_r4az.sh
# round 4, call az: final validation on the final sources: smoke, whole gpu suite, fuzz campaigns, the round's profiles, default bench
mkdir -p gpurun_out
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep "smoke" | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
CSR5_FUZZ_SEED=4242 CSR5_FUZZ_CASES=3000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
ROUND=r04 timeout 2400 bash scripts/gpu_profiles_round.sh 2>&1 | grep -E "^rmat|^webbase|^scircuit|^nd24k" | cut -c1-200
timeout 600 python bench.py > gpurun_out/r4y_bench.json 2> gpurun_out/r4y_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4y_bench.json'))
r=d['roofline']; print('headline', d['value'], 'GFLOPS', d['ms_per_step'], 'ms', 'frac', r['frac'], 'traffic', r.get('traffic'), r.get('traffic_source'), 'live', (r.get('x_live') or {}).get('launch_us'), 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'), (r.get('narrowed_values') or {}).get('y_bit_identical_to_headline_run'))
for c in d.get('configs', []):
    rr=c.get('roofline', {}); print(c.get('workload','')[:30], c.get('value'), rr.get('frac'), rr.get('launch_us'), (rr.get('warm') or {}).get('frac'), rr.get('traffic'))
print(d.get('cpu_baseline'))
PY
) }

r4b() { ( # ---- round4/gpu_r4b.sh
# round 4, call b: what holds the L1 (TCP) of k_spmv_range -- request latency, TLB, FIFOs (one PMC pass each)
export KFILTER=k_spmv_range
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" bash scripts/gpu_pmc1.sh tlb
PMC="TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_TCR_RDRET_STALL_sum TCP_TD_TCP_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh fifo
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_CACHE_MISS_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" bash scripts/gpu_pmc1.sh l1
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh ta
) }

r4ba() { ( # ---- round4/gpu_r4ba.sh
# round 4, call ba: did the storage-type template parameter change the plain fp64 kernel?  previous commit vs current, same call
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for v in prev cur prev cur; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w; done; done
) }

r4bd() { ( # ---- round4/gpu_r4bd.sh
# round 4, call bd: R-MAT 25 / 26 again with the graph instantiated before the timed call
timeout 900 python scripts/experiments/scale_check.py --scale 25 2>&1 | tail -1
timeout 1500 python scripts/experiments/scale_check.py --scale 26 2>&1 | tail -1
) }

r4be() { ( # ---- round4/gpu_r4be.sh
# round 4, call be: long fuzz campaigns on the final sources (narrow-values and x-snapshot modes drawn per case)
for seed in 90001 90002 90003; do
  CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=4000 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror|assert" | tail -2
done
) }

r4bf() { ( # ---- round4/gpu_r4bf.sh
# round 4, call bf: the small power-law configs with a forced hot table on the round-4 range kernel (round 3 measured them on its kernel)
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {}); c = d['config']
        print('%-22s slabs %2s hot %d/%2d%% cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (c['workload'][:22], c.get('column_slabs'), int(c.get('slab_hot_table', 0)), c.get('slab_hot_cover_pct', 0), r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in webbase scircuit; do
  python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line
  for s in 8 16; do
    python bench.py --no-cpu-baseline --no-sub-configs --workload $w --slabs $s --slab-hot force 2>&1 | tail -1 | line
  done
done
) }

r4bg() { ( # ---- round4/gpu_r4bg.sh
# round 4, call bg: y_offset prefix of the range kernel by six DPP adds instead of six __shfl_up (ds_bpermute) steps; parity, then same-call pairs
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
CSR5_FUZZ_SEED=515 CSR5_FUZZ_CASES=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs --no-side-figures "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for v in base dppscan base dppscan base dppscan; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w; done; done
) }

r4bh() { ( # ---- round4/gpu_r4bh.sh
# round 4, call bh: the combine's loads of P (and of the row bytes) with the non-temporal hint -- P is dead after the combine
one() { python bench.py --no-cpu-baseline --no-sub-configs --no-side-figures "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22 webbase; do for v in base abl_combine_nt abl_combine_ntP base abl_combine_nt abl_combine_ntP; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w --no-cold; done; done
) }

r4bi() { ( # ---- round4/gpu_r4bi.sh
# round 4, call bi: compact column codes of the fp64 hot child (16 bits + one byte per COLD element): parity, then same-call pairs
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep "smoke" | tail -3
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
CSR5_FUZZ_SEED=616 CSR5_FUZZ_CASES=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['value'], r['launch_us'], r['frac'], 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'), 'conv', d['config'].get('csr_to_csr5_ms'))"; }
for w in rmat24 rmat22; do for v in base cx base cx base cx; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w; done; done
) }

r4bj() { ( # ---- round4/gpu_r4bj.sh
# round 4, call bj: one sequential stream per wavefront (6-KB tile records: column words, then values) instead of two arrays -- does
# the HBM side care how many concurrent streams the 2 048 wavefronts read?
cd scripts/probes
echo "## two arrays"; timeout 120 ./lds_dma_streams_w0 268435456 28,0 | grep -v "LDS-DMA\|consumers"
echo "## one array of tile records"; timeout 120 ./lds_dma_streams_il 268435456 28,0
echo "## two arrays"; timeout 120 ./lds_dma_streams_w0 268435456 28 | grep -v "LDS-DMA\|consumers"
echo "## one array of tile records"; timeout 120 ./lds_dma_streams_il 268435456 28
) }

r4bk() { ( # ---- round4/gpu_r4bk.sh
# round 4, call bk: how much does the size of the region the cold gathers fall into matter (per XCD: 3 600 KB = the product's, down to L1-sized)?
cd scripts/probes
for kb in 3600 1024 256 64 16; do timeout 120 ./lds_dma_streams_w0 268435456 28 $kb | grep "^##\|registers + gathers\|no streams"; done
) }

r4bl() { ( # ---- round4/gpu_r4bl.sh
# round 4, call bl: the value stream as four 16-byte loads per lane instead of eight 8-byte ones (half the vector-memory instructions, same lines)
cd scripts/probes
for r in 1 2; do
echo "## 8 x 8 bytes"; timeout 120 ./lds_dma_streams_w0 268435456 28 | grep "registers + gathers\|streams only"
echo "## 4 x 16 bytes"; timeout 120 ./lds_dma_streams_wv 268435456 28 | grep "registers + gathers\|streams only"
done
) }

r4bm() { ( # ---- round4/gpu_r4bm.sh
# round 4, call bm: the hot child's values in lane-major 16-byte pieces (half the value load instructions): parity, then same-call pairs
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep "smoke" | tail -3
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
CSR5_FUZZ_SEED=717 CSR5_FUZZ_CASES=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['value'], r['launch_us'], r['frac'], 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'), 'conv', d['config'].get('csr_to_csr5_ms'))"; }
for w in rmat24 rmat22; do for v in base pieces base pieces base pieces; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w; done; done
) }

r4bn() { ( # ---- round4/gpu_r4bn.sh
# round 4, call bn: lane-major 16-byte pieces of the hot child's values, more same-call pairs on R-MAT 24 (headline and narrowed)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['value'], r['launch_us'], r['frac'], 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'))"; }
for v in base pieces base pieces base pieces base pieces base pieces; do echo -n "rmat24 $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload rmat24; done
) }

r4bo() { ( # ---- round4/gpu_r4bo.sh
# round 4, call bo: the fp32 copy of the values (CSR5HIP_OPT_NARROW_VALUES) in lane-major pieces of four floats: parity, then same-call pairs
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -k "narrow or hot" 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
CSR5_FUZZ_SEED=818 CSR5_FUZZ_CASES=1000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['value'], r['launch_us'], r['frac'], 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'))"; }
for v in base pieces base pieces base pieces base pieces; do echo -n "rmat24 $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload rmat24; done
for v in base pieces base pieces; do echo -n "rmat22 $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload rmat22; done
) }

r4bp() { ( # ---- round4/gpu_r4bp.sh
# round 4, call bp: TIMING ONLY -- k_spmv's column / value streams fetched with 16-byte loads (elements in the wrong lanes: wrong results);
# is the one-tile kernel of the small configs sensitive to the number of its load instructions?
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {}); c = d['config']
        print('%-22s sigma %2d cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (c['workload'][:22], c['sigma'], r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in nd24k "nd24k --dtype f64"; do for v in base abl_widespmv base abl_widespmv; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line; done; done
for v in base abl_widespmv; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload webbase --sigma 8 2>&1 | tail -1 | line; done
for v in base abl_widespmv; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload scircuit --sigma 8 2>&1 | tail -1 | line; done
) }

r4bq() { ( # ---- round4/gpu_r4bq.sh
# round 4, call bq: TIMING ONLY -- the tile's cold lanes issued as ceil(cold / 64) FULL gather instructions instead of 8 quarter-full ones
# (what a cross-lane compaction would issue; its own cost is not in here)
cd scripts/probes
for r in 1 2; do
echo "## 8 gather instructions, 28 % of the lanes cold"; timeout 120 ./lds_dma_streams_wv 268435456 28 | grep "registers + gathers\|no streams"
echo "## compacted gather instructions"; timeout 120 ./lds_dma_streams_cg 268435456 28 | grep "registers + gathers\|no streams"
done
) }

r4br() { ( # ---- round4/gpu_r4br.sh
# round 4, call br: the x-window of the one-tile kernel staged with 16-byte loads / LDS stores (4 + 4 instead of 16 + 16 per tile): parity, pairs
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {}); c = d['config']
        print('%-22s %s xwin %d cover %3d%% cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (c['workload'][:22], d['dtype'], int(c.get('lds_x_window', 0)), c.get('x_window_cover_pct', 0), r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in nd24k "nd24k --dtype f64"; do for v in base xwide base xwide base xwide; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line; done; done
) }

r4c() { ( # ---- round4/gpu_r4c.sh
# round 4, call c: same-call A/B of issue-order / scalar tile_ptr variants of k_spmv_range, and child sigma 4
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w base"; one --workload $w
    echo "== $w tpscalar"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_tpscalar.so one --workload $w
    echo "== $w streams-first"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_sfirst.so one --workload $w
  done
done
echo "== rmat24 sigma 4"; one --workload rmat24 --sigma 4
echo "== rmat22 sigma 4"; one --workload rmat22 --sigma 4
) }

r4d() { ( # ---- round4/gpu_r4d.sh
# round 4, call d: parity after the mark-based cold ranking, RCCL one-device test, A/B vs round 3, conversion timeline
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w r03"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_r03.so one --workload $w
    echo "== $w new"; one --workload $w
  done
done
bash scripts/gpu_convtrace.sh rmat24 62 | tail -34
) }

r4e() { ( # ---- round4/gpu_r4e.sh
# round 4, call e: sigma sweep on the round-4 kernels (fp32 table), cold knobs of the small configs
mkdir -p gpurun_out
cold() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   cold %.2f us frac %.3f | warm %.2f us frac %.3f | sigma %d' % (r['launch_us'], r['frac'], r['warm']['launch_us'], r['warm']['frac'], d['config']['sigma']))"; }
for w in nd24k; do
  echo "== $w default"; cold --workload $w
  echo "== $w nt force"; cold --workload $w --stream-nt force
  echo "== $w sigma 32"; cold --workload $w --sigma 32
  echo "== $w sigma 24"; cold --workload $w --sigma 24
  echo "== $w sigma 12"; cold --workload $w --sigma 12
  echo "== $w lds-y force"; cold --workload $w --lds-y force
done
for w in scircuit webbase; do
  echo "== $w default"; cold --workload $w
  echo "== $w nt force"; cold --workload $w --stream-nt force
  echo "== $w sigma 8"; cold --workload $w --sigma 8
  echo "== $w sigma 16"; cold --workload $w --sigma 16
done
timeout 1500 python scripts/experiments/sigma_table.py > gpurun_out/r04_sigma_table.txt 2>&1; tail -3 gpurun_out/r04_sigma_table.txt
) }

r4f() { ( # ---- round4/gpu_r4f.sh
# round 4, call f: parity after the k_range_finish rewrite; fp64 sigma 10 vs 16 on nd24k-like fp64 and R-MAT
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
cold() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   cold %.2f us frac %.3f | warm %.2f us frac %.3f | sigma %d xwin %s' % (r['launch_us'], r['frac'], r['warm']['launch_us'], r['warm']['frac'], d['config']['sigma'], d['config']['lds_x_window']))"; }
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for s in -1 10 12 8; do echo "== nd24k f64 sigma $s"; cold --workload nd24k --dtype f64 --sigma $s; done
for s in -1 10; do echo "== rmat22 sigma $s"; one --workload rmat22 --sigma $s; echo "== rmat24 sigma $s"; one --workload rmat24 --sigma $s; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 3 > /dev/null 2>&1; f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); grep -E "k_spmv_range|k_range_finish|k_slab_combine|k_x_permute" $f | cut -c1-200
) }

r4g() { ( # ---- round4/gpu_r4g.sh
# round 4, call g: row blocks of R-MAT 24 alone (8 blocks: ranks 0 3 7; slabs auto vs 16), whole matrix first
timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --slabs 16
) }

r4h() { ( # ---- round4/gpu_r4h.sh
# round 4, call h: the whole gpu suite, then the default bench line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 python bench.py > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err; tail -c 600 gpurun_out/r4h_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4h_bench.json'))
r=d['roofline']; print('headline', d['value'], 'GFLOPS', d['ms_per_step'], 'ms', 'frac', r['frac'], 'live', r.get('x_live'))
for c in d.get('configs', []):
    rr=c.get('roofline', {}); print(c.get('workload','')[:30], c.get('value'), rr.get('frac'), rr.get('launch_us'), (rr.get('warm') or {}).get('frac'))
print(d.get('cpu_baseline'))
PY
) }

r4i() { ( # ---- round4/gpu_r4i.sh
# round 4, call i: tiles per wavefront of the x-window kernel (nd24k-like fp32 and fp64), warm and cold
cold() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   cold %.2f us frac %.3f | warm %.2f us frac %.3f | sigma %d xwin %s tpw %s' % (r['launch_us'], r['frac'], r['warm']['launch_us'], r['warm']['frac'], d['config']['sigma'], d['config']['lds_x_window'], d['config'].get('tiles_per_wave')))"; }
for rep in 1 2; do
for t in 1 2 3 4; do echo "== nd24k f32 tpw $t"; cold --workload nd24k --tiles-per-wave $t; done
done
for t in 1 2 4; do echo "== nd24k f64 tpw $t"; cold --workload nd24k --dtype f64 --tiles-per-wave $t; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "window or xwin or nd24k or zoo" 2>&1 | tail -3
) }

r4j() { ( # ---- round4/gpu_r4j.sh
# round 4, call j: wavefronts per CU of the persistent kernel vs table size (16 waves / 12 288 slots, 12 / 14 336, 8 / 16 384)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 16 waves"; one --workload $w
    echo "== $w 12 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w12.so one --workload $w
    echo "== $w 8 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8.so one --workload $w
  done
done
) }

r4k() { ( # ---- round4/gpu_r4k.sh
# round 4, call k: fewer wavefronts per CU, larger table (8 waves / 16 384 slots, 6 / 17 408, 4 / 18 432)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 8 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8.so one --workload $w
    echo "== $w 6 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w6.so one --workload $w
    echo "== $w 4 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w4.so one --workload $w
  done
done
) }

r4l() { ( # ---- round4/gpu_r4l.sh
# round 4, call l: prefetch depth 3 (streams two tiles ahead) at 8 and 16 wavefronts per CU
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 8 waves depth 2"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8.so one --workload $w
    echo "== $w 8 waves depth 3"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8d3.so one --workload $w
    echo "== $w 16 waves depth 3"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16d3.so one --workload $w
  done
done
) }

r4m() { ( # ---- round4/gpu_r4m.sh
# round 4, call m: parity with 8 wavefronts per CU, blocks of R-MAT 24 alone, small configs unchanged?
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --slabs 8
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 4 --ranks 0,3
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 2 --ranks 0,1
) }

r4n() { ( # ---- round4/gpu_r4n.sh
# round 4, call n: slab count with the 16 384-slot table (8 / 16 / 32), R-MAT 24 and 22
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for s in 16 32 8; do echo "== rmat24 slabs $s"; one --workload rmat24 --slabs $s; done
  for s in 8 16 32; do echo "== rmat22 slabs $s"; one --workload rmat22 --slabs $s; done
done
) }

r4o() { ( # ---- round4/gpu_r4o.sh
# round 4, call o: combine whose later rounds load only the runs that are that long (A/B vs the previous library)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_cmb0.so one --workload $w
    echo "== $w after"; one --workload $w
  done
done
timeout 600 python -m pytest tests/test_gpu_slabs.py -x -q 2>&1 | tail -2
) }

r4p() { ( # ---- round4/gpu_r4p.sh
# round 4, call p: windowed y-compaction: region 4096 B (whole tile, table 16 384 slots) / 2048 / 1024 / 512 B per wavefront
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w region 4096"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wl4096.so one --workload $w
    echo "== $w region 2048"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wl2048.so one --workload $w
    echo "== $w region 1024"; one --workload $w
    echo "== $w region 512"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wl512.so one --workload $w
  done
done
) }

r4q() { ( # ---- round4/gpu_r4q.sh
# round 4, call q: how often must a column be used to earn a table slot?  (blocks of R-MAT 24 alone: their columns are used 1/8 as often)
for u in 48 16 4; do
  echo "== min uses $u, auto slabs"; CSR5_EXPERIMENT_HOT_MIN_USES=$u timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7
  echo "== min uses $u, 16 slabs"; CSR5_EXPERIMENT_HOT_MIN_USES=$u timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --slabs 16
done
echo "== whole matrix, min uses 48 / 4"; CSR5_EXPERIMENT_HOT_MIN_USES=48 timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
CSR5_EXPERIMENT_HOT_MIN_USES=4 timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
) }

r4r() { ( # ---- round4/gpu_r4r.sh
# round 4, call r: all row blocks of R-MAT 24 alone, row weight 2 (default) and 3; whole matrix; R-MAT 22 / 20 sanity
mkdir -p gpurun_out
{
echo "## scripts/experiments/shard_alone.py: every row block of the strong-scaling R-MAT 24 ALONE on one MI355X (cost balance nnz + 2*rows), round-4 kernels, x snapshot"
timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 2 --ranks 0,1
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 4 --ranks 0,1,2,3
timeout 1200 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,1,2,3,4,5,6,7
echo "## row weight 3"
timeout 1200 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,1,2,3,4,5,6,7 --row-weight 3
echo "## row weight 1"
timeout 1200 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --row-weight 1
} 2>/dev/null | tee gpurun_out/r04_shards.txt
) }

r4s() { ( # ---- round4/gpu_r4s.sh
# round 4, call s: final row-block table (all 15 lines, same box) + the round's profiles
bash scripts/experiments/round4/gpu_r4r.sh > /dev/null 2>&1
cat gpurun_out/r04_shards.txt | cut -c1-120 | tail -32
ROUND=r04 timeout 2400 bash scripts/gpu_profiles_round.sh 2>&1 | tail -12
) }

r4t() { ( # ---- round4/gpu_r4t.sh
# round 4, call t: PMC passes on the final k_spmv_range and k_slab_combine (what holds the L1: request latency, TLB, FIFOs, L2)
mkdir -p gpurun_out
{
for K in k_spmv_range "k_slab_combine<double, 16"; do
export KFILTER="$K"
echo "#### $K"
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat --no-sub-configs --no-side-figures
PMC="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCC_WRITE_REQ_sum" bash scripts/gpu_pmc1.sh tlb --no-sub-configs --no-side-figures
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2 --no-sub-configs --no-side-figures
PMC="TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TOTAL_ACCESSES_sum" bash scripts/gpu_pmc1.sh l1 --no-sub-configs --no-side-figures
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" bash scripts/gpu_pmc1.sh ta --no-sub-configs --no-side-figures
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq --no-sub-configs --no-side-figures
rm -rf gpurun_out/pmc1_*
done
} 2>&1 | grep -v "^$" > gpurun_out/r04_pmc_raw.txt
cat gpurun_out/r04_pmc_raw.txt
) }

r4u() { ( # ---- round4/gpu_r4u.sh
# round 4, call u: PMC passes on k_slab_combine<double, 16> of R-MAT 24 only
mkdir -p gpurun_out
export KFILTER="k_slab_combine<double, 16"
{
echo "#### $KFILTER"
PMC="TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_READ_sum" bash scripts/gpu_pmc1.sh lat --no-sub-configs
PMC="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" bash scripts/gpu_pmc1.sh l2 --no-sub-configs
PMC="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" bash scripts/gpu_pmc1.sh ta --no-sub-configs
PMC="SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" bash scripts/gpu_pmc1.sh sq --no-sub-configs
rm -rf gpurun_out/pmc1_*
} 2>&1 | grep -v "^$" > gpurun_out/r04_pmc_combine_raw.txt
cat gpurun_out/r04_pmc_combine_raw.txt
) }

r4v() { ( # ---- round4/gpu_r4v.sh
# round 4, call v: combine with packed row bytes in its first round (A/B vs the previous library), parity
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22 webbase; do
    echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_cmb0.so one --workload $w
    echo "== $w after"; one --workload $w
  done
done
) }

r4w() { ( # ---- round4/gpu_r4w.sh
# round 4, call w: fuzz campaign on the final kernels (new seeds) + R-MAT 25 / 26 beyond the BASELINE size
for seed in 401 402 403 404 405; do
  CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=3000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k seeded_fuzz 2>&1 | tail -1
done
for seed in 501 502 503; do
  CSR5_FUZZ_SCALE=30 CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=600 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k seeded_fuzz 2>&1 | tail -1
done
timeout 900 python scripts/experiments/scale_check.py --scale 25 2>&1 | tail -2
timeout 1200 python scripts/experiments/scale_check.py --scale 26 2>&1 | tail -2
) }

r4x() { ( # ---- round4/gpu_r4x.sh
# round 4, call x: does a denser column sample (1 chunk in 16 / 8 instead of 64) order the cold regions better?  traffic + time + conversion
for cap in 64 16 8; do
  echo "== stride cap $cap"
  export CSR5_EXPERIMENT_STRIDE_CAP=$cap
  python bench.py --no-cpu-baseline --no-sub-configs 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230
  PMC="FETCH_SIZE" KFILTER=k_ bash scripts/gpu_pmc1.sh f$cap --no-sub-configs | grep -v "^$"
  PMC="WRITE_SIZE" KFILTER=k_ bash scripts/gpu_pmc1.sh w$cap --no-sub-configs | grep -v "^$"
  PMC="TCC_MISS_sum TCC_HIT_sum" KFILTER=k_spmv_range bash scripts/gpu_pmc1.sh m$cap --no-sub-configs | grep -v "^$"
  rm -rf gpurun_out/pmc1_*
done
) }

r4y() { ( # ---- round4/gpu_r4y.sh
# round 4, call y: smoke, the whole gpu suite, the round's profiles on the final sources, default bench
mkdir -p gpurun_out
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ROUND=r04 timeout 2400 bash scripts/gpu_profiles_round.sh 2>&1 | grep -E "^rmat|^webbase|^scircuit|^nd24k" | cut -c1-200
timeout 600 python bench.py > gpurun_out/r4y_bench.json 2> gpurun_out/r4y_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4y_bench.json'))
r=d['roofline']; print('headline', d['value'], 'GFLOPS', d['ms_per_step'], 'ms', 'frac', r['frac'], 'traffic', r.get('traffic'), r.get('traffic_source'), 'live', (r.get('x_live') or {}).get('launch_us'))
for c in d.get('configs', []):
    rr=c.get('roofline', {}); print(c.get('workload','')[:30], c.get('value'), rr.get('frac'), rr.get('launch_us'), (rr.get('warm') or {}).get('frac'), rr.get('traffic'))
print(d.get('cpu_baseline'))
PY
) }

DESCRIPTIONS=$(cat <<'EOT'
r3_coldpaths	# cold-protocol figures of the small configs under each path (which path should the auto rule pick when the working set is cold?)
r3_final	# last call of the round: full -m gpu suite, smoke, then the round's profiles and the default bench line on the final sources
r3a	# round 3, call A: correctness of the range kernel, then same-call A/B against the round-2 library
r3b	# round 3, call B: parallel finish kernel + run-streaming combine: tests, per-kernel times by slab count, all-hot floor
r3d	#
r3e	#
r3f	#
r3h	#
r3j	#
r3k	#
r3l	#
r3n	#
r3o	#
r3p	#
r3q	#
r3r	#
r3s	#
r3t	#
r3w	#
r3x	#
r3y	# packed column codes: full GPU suite, A/B against the previous build, conversion trace
r3z	# every row block of the strong-scaling R-MAT 24 ALONE on the one GPU, round-3 kernels (estimate of the N-GPU step: the slowest block)
r4a	# round 4, call a: parity of the permuted-x path, then same-call A/B against the round-3 library
r4ab	# round 4, call ab: where does the final tile kernel's time go?  builds of the product with one piece of work removed
r4ac	# round 4, call ac: child sigma 8 / 12 / 16 at 8 wavefronts per CU (a larger tile amortises the per-tile work that two wavefronts
r4ad	# round 4, call ad: fp64 child sigma 16 / 12 / 8 with the SAME 16 384-slot table (4-KB y region, a tile of very short rows walks its
r4ae	# round 4, call ae: slab hash granule (columns hashed together) now that cold gathers read the dense permuted copy, not lines of x
r4af	# round 4, call af: 16 (12) wavefronts per CU taking turns on 8 (6) y-compaction regions (LDS lock): occupancy of the round-3
r4ak	# round 4, call ak: the child's bit flags ride in bit 22 of the column codes, y_offset is recomputed in the kernel, the descriptor
r4al	# round 4, call al: whole gpu suite after dropping the hot child's descriptor array; conversion time; bench
r4am	# round 4, call am: k_spmv touches the streams of the tile one resident set ahead (CSR5_PREFETCH = percent of a resident set);
r4an	# round 4, call an: the stream / gather probe again with its modes repaired (round 4's first version computed the column codes of the
r4ao	# round 4, call ao: the combine as a stream (persistent wavefronts, two row blocks in flight each); CSR5_COMBINE_STREAM = workgroups per CU
r4ap	# round 4, call ap: the combine's LDS read-modify-write chain (16 dependent ds_read / add / ds_write rounds) as ds_add_f64 without return
r4aq	# round 4, call aq: what holds k_slab_combine (R-MAT 24)?  LDS conflicts, the dispatcher's resource stalls, instruction classes
r4ar	# round 4, call ar: workgroup size of k_slab_combine (one row block per wavefront): 64 / 128 / 256 (product) / 512 / 1024 threads
r4as	# round 4, call as: TIMING of k_slab_combine if a row block's S runs of partials lay one behind the other in P (block-major P instead of
r4at	# round 4, call at: ablations of k_slab_combine on R-MAT 24 (wrong results by design): 3 = no partial / row-byte loads, 6 = no partial
r4au	# round 4, call au: the combine as a stream, second form (unit of work = one round of a block, next round's loads issued first, ds_add_f64)
r4av	# round 4, call av: k_slab_combine with 32-bit byte offsets on a scalar base (288 -> 256 vector instructions per block)
r4aw	# round 4, call aw: per-kernel times of row blocks 1 and 3 of 8 (R-MAT 24) alone on the GPU: where do the blocks' 170-178 us go?
r4ax	# round 4, call ax: gaps between the three kernels of a step inside the replayed graph (row block 3 of 8, and the whole matrix)
r4ay	# round 4, call ay: CSR5HIP_OPT_NARROW_VALUES (fp32-exact fp64 values streamed as fp32): parity and the side figure of the bench
r4az	# round 4, call az: final validation on the final sources: smoke, whole gpu suite, fuzz campaigns, the round's profiles, default bench
r4b	# round 4, call b: what holds the L1 (TCP) of k_spmv_range -- request latency, TLB, FIFOs (one PMC pass each)
r4ba	# round 4, call ba: did the storage-type template parameter change the plain fp64 kernel?  previous commit vs current, same call
r4bd	# round 4, call bd: R-MAT 25 / 26 again with the graph instantiated before the timed call
r4be	# round 4, call be: long fuzz campaigns on the final sources (narrow-values and x-snapshot modes drawn per case)
r4bf	# round 4, call bf: the small power-law configs with a forced hot table on the round-4 range kernel (round 3 measured them on its kernel)
r4bg	# round 4, call bg: y_offset prefix of the range kernel by six DPP adds instead of six __shfl_up (ds_bpermute) steps; parity, then same-call pairs
r4bh	# round 4, call bh: the combine's loads of P (and of the row bytes) with the non-temporal hint -- P is dead after the combine
r4bi	# round 4, call bi: compact column codes of the fp64 hot child (16 bits + one byte per COLD element): parity, then same-call pairs
r4bj	# round 4, call bj: one sequential stream per wavefront (6-KB tile records: column words, then values) instead of two arrays -- does
r4bk	# round 4, call bk: how much does the size of the region the cold gathers fall into matter (per XCD: 3 600 KB = the product's, down to L1-sized)?
r4bl	# round 4, call bl: the value stream as four 16-byte loads per lane instead of eight 8-byte ones (half the vector-memory instructions, same lines)
r4bm	# round 4, call bm: the hot child's values in lane-major 16-byte pieces (half the value load instructions): parity, then same-call pairs
r4bn	# round 4, call bn: lane-major 16-byte pieces of the hot child's values, more same-call pairs on R-MAT 24 (headline and narrowed)
r4bo	# round 4, call bo: the fp32 copy of the values (CSR5HIP_OPT_NARROW_VALUES) in lane-major pieces of four floats: parity, then same-call pairs
r4bp	# round 4, call bp: TIMING ONLY -- k_spmv's column / value streams fetched with 16-byte loads (elements in the wrong lanes: wrong results);
r4bq	# round 4, call bq: TIMING ONLY -- the tile's cold lanes issued as ceil(cold / 64) FULL gather instructions instead of 8 quarter-full ones
r4br	# round 4, call br: the x-window of the one-tile kernel staged with 16-byte loads / LDS stores (4 + 4 instead of 16 + 16 per tile): parity, pairs
r4c	# round 4, call c: same-call A/B of issue-order / scalar tile_ptr variants of k_spmv_range, and child sigma 4
r4d	# round 4, call d: parity after the mark-based cold ranking, RCCL one-device test, A/B vs round 3, conversion timeline
r4e	# round 4, call e: sigma sweep on the round-4 kernels (fp32 table), cold knobs of the small configs
r4f	# round 4, call f: parity after the k_range_finish rewrite; fp64 sigma 10 vs 16 on nd24k-like fp64 and R-MAT
r4g	# round 4, call g: row blocks of R-MAT 24 alone (8 blocks: ranks 0 3 7; slabs auto vs 16), whole matrix first
r4h	# round 4, call h: the whole gpu suite, then the default bench line
r4i	# round 4, call i: tiles per wavefront of the x-window kernel (nd24k-like fp32 and fp64), warm and cold
r4j	# round 4, call j: wavefronts per CU of the persistent kernel vs table size (16 waves / 12 288 slots, 12 / 14 336, 8 / 16 384)
r4k	# round 4, call k: fewer wavefronts per CU, larger table (8 waves / 16 384 slots, 6 / 17 408, 4 / 18 432)
r4l	# round 4, call l: prefetch depth 3 (streams two tiles ahead) at 8 and 16 wavefronts per CU
r4m	# round 4, call m: parity with 8 wavefronts per CU, blocks of R-MAT 24 alone, small configs unchanged?
r4n	# round 4, call n: slab count with the 16 384-slot table (8 / 16 / 32), R-MAT 24 and 22
r4o	# round 4, call o: combine whose later rounds load only the runs that are that long (A/B vs the previous library)
r4p	# round 4, call p: windowed y-compaction: region 4096 B (whole tile, table 16 384 slots) / 2048 / 1024 / 512 B per wavefront
r4q	# round 4, call q: how often must a column be used to earn a table slot?  (blocks of R-MAT 24 alone: their columns are used 1/8 as often)
r4r	# round 4, call r: all row blocks of R-MAT 24 alone, row weight 2 (default) and 3; whole matrix; R-MAT 22 / 20 sanity
r4s	# round 4, call s: final row-block table (all 15 lines, same box) + the round's profiles
r4t	# round 4, call t: PMC passes on the final k_spmv_range and k_slab_combine (what holds the L1: request latency, TLB, FIFOs, L2)
r4u	# round 4, call u: PMC passes on k_slab_combine<double, 16> of R-MAT 24 only
r4v	# round 4, call v: combine with packed row bytes in its first round (A/B vs the previous library), parity
r4w	# round 4, call w: fuzz campaign on the final kernels (new seeds) + R-MAT 25 / 26 beyond the BASELINE size
r4x	# round 4, call x: does a denser column sample (1 chunk in 16 / 8 instead of 64) order the cold regions better?  traffic + time + conversion
r4y	# round 4, call y: smoke, the whole gpu suite, the round's profiles on the final sources, default bench
EOT
)

case "${1:-list}" in
  list) echo "$DESCRIPTIONS" ;;
  call) shift; "$@" ;;
  *) echo "usage: $0 list | call <name>"; exit 2 ;;
esac
